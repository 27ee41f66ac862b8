#!/bin/bash
# One-shot probe of the GPU box for any OpenCV: prints what it finds; never fails.
out=${1:-gpurun_out/r06/opencv_probe.txt}
mkdir -p "$(dirname "$out")"
{
  echo "== python3 -c 'import cv2'"
  python3 -c "import cv2; print(cv2.__version__, cv2.__file__)" 2>&1 | tail -1
  echo "== pkg-config"
  pkg-config --modversion opencv4 2>&1; pkg-config --modversion opencv 2>&1
  echo "== find libopencv_imgproc* / opencv2 headers / cv2*.so"
  find / -xdev \( -name 'libopencv_imgproc*' -o -name 'libopencv_core*' -o -name 'cv2*.so' -o -path '*/opencv2/core.hpp' -o -path '*/opencv2/opencv.hpp' \) 2>/dev/null | head -40
  echo "== pip list | grep -i opencv"
  python3 -m pip list 2>/dev/null | grep -i -E 'opencv|cv2' || echo none
  echo "== ldconfig -p | grep opencv"
  ldconfig -p 2>/dev/null | grep -i opencv || echo none
  echo "== wheelhouse"
  ls /opt/wheelhouse 2>/dev/null | grep -i -E 'opencv|cv' || echo none
  echo "== glibc / gcc"
  ldd --version | head -1; gcc --version | head -1
  echo "== nproc"; nproc
} > "$out" 2>&1
cat "$out"
exit 0
