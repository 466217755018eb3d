#!/bin/bash
# measurement-only builds of the library with -DAQC_ABL=<bits> / other -D switches (see aqc_fast.hpp): build/ablate/lib_<tag>.so
# usage: tools/build_ablate.sh tag1:"-DAQC_ABL=8" tag2:"-DAQC_ABL=16 -DFOO=1" ...
cd "$(dirname "$0")/.."; mkdir -p build/ablate
python __graft_entry__.py build > /dev/null || exit 1      # (the host objects)
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-function -Wno-pass-failed -I include $flags \
    afterqc_amd/csrc/aqc_capi.hip -Wl,build/aqc_pipe.o -Wl,build/aqc_inflate.o -Wl,build/aqc_gunzip.o -Wl,build/aqc_deflate.o -lz -lpthread -ldl -o build/ablate/lib_$tag.so &
done
wait
ls -la build/ablate
