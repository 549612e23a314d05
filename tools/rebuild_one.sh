#!/bin/bash
# tools/rebuild_one.sh <file.hip|file.c> ... -- dev loop: compile only the named translation units, mark the rest up to date, relink librfx.so
set -e
D=/root/repo/rayforce_amd/csrc
cd $D
for f in "$@"; do
  b=$(basename $f); o=build/${b%.*}.o
  case $b in
    *.hip) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed -fno-fast-math -ffp-contract=off -I../../include $EXTRA -c $b -o $o ;;
    *.c) gcc -O2 -std=gnu11 -fPIC -Wall -Wextra -I../../include -c $b -o $o ;;
  esac
done
make -s -t > /dev/null 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librfx.so build/*.o -Wl,--no-undefined -ldl -lpthread
ls -la --time-style=+%H:%M ../librfx.so
