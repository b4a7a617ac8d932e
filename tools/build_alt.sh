#!/bin/bash
# Build a VARIANT of libpixelhip.so into tools/alt/<name>/libpixelhip.so: a copy of csrc/ with `constexpr int NAME = <old>;` rewritten to the
# given values (kernel-level switches that must be compile-time constants), for A/B runs with PXL_LIB_PATH=tools/alt/<name>/libpixelhip.so.
#   tools/build_alt.sh a_nt DMA_AUX_A=2
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=$PWD/tools/alt/$name; tmp=$(mktemp -d); mkdir -p $out $tmp/pixelssl_amd/csrc $tmp/include
cp pixelssl_amd/csrc/*.hip pixelssl_amd/csrc/*.cpp pixelssl_amd/csrc/*.h $tmp/pixelssl_amd/csrc/; cp include/*.h $tmp/include/
for kv in "$@"; do
  k=${kv%%=*}; v=${kv#*=}
  grep -l "constexpr int $k = " $tmp/pixelssl_amd/csrc/* | xargs sed -i "s/constexpr int $k = [-0-9]*;/constexpr int $k = $v;/"
  grep -h "constexpr int $k = " $tmp/pixelssl_amd/csrc/*
done
ls $tmp/pixelssl_amd/csrc/*.hip $tmp/pixelssl_amd/csrc/*.cpp | xargs -P 8 -I{} /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -x hip -c {} -o {}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libpixelhip.so $tmp/pixelssl_amd/csrc/*.o
rm -rf $tmp
ls -la $out/libpixelhip.so
