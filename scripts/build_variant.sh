#!/bin/bash
# Dev tool: build a variant of liblcsgpu.so whose lcs_kernels.hip is compiled with extra -D flags (A/B runs on the
# GPU box: LCSGPU_LIB=famsa_amd/_variants/<name>/liblcsgpu.so python scripts/...).  The other objects are reused.
#   scripts/build_variant.sh <name> [-DLCS_SEGW=16 ...]
set -e
name=$1; shift
here=$(cd "$(dirname "$0")/.." && pwd)
src=$here/famsa_amd/csrc
out=$here/famsa_amd/_variants/$name
mkdir -p $out
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Xclang -target-feature -Xclang -load-store-opt $*"
LLVM=/opt/rocm/lib/llvm/bin
/opt/rocm/bin/hipcc $FLAGS --cuda-device-only -S -o $out/lcs_kernels.dev.s $src/lcs_kernels.hip 2>&1 | grep -v "not a recognized" || true
python3 $src/recolor_vgprs.py $out/lcs_kernels.dev.s $out/lcs_kernels.rec.s --only 'lcs_rows_kernel_pipe|lcs_long_kernel' --report | tail -2
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $out/lcs_kernels.rec.s -o $out/lcs_kernels.dev.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $out/lcs_kernels.hsaco $out/lcs_kernels.dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
    -input=/dev/null -input=$out/lcs_kernels.hsaco -output=$out/lcs_kernels.hipfb
/opt/rocm/bin/hipcc $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $out/lcs_kernels.hipfb -c -o $out/lcs_kernels.o $src/lcs_kernels.hip 2>&1 | grep -v "not a recognized" || true
objs=""
for n in lcsgpu_api lcsgpu_trees lcsgpu_fasttree tree_kernels mst_kernels clarans_kernels upload_kernels; do objs="$objs $src/_obj/$n.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/liblcsgpu.so $objs $out/lcs_kernels.o
cp $out/lcs_kernels.rec.s /tmp/variant_$name.s; rm -f $out/*.s $out/*.o $out/*.hsaco $out/*.hipfb
ls -la $out/liblcsgpu.so
