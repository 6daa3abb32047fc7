#!/bin/bash
# Fixed cost of a famsa-gpu run on small inputs (BASELINE config C2): wall time of the whole process next to
# the tool's own stage timers, and what a bare HIP program pays for the runtime alone.  Dev/measurement tool.
cd "$(dirname "$0")/.."
H=tests/golden/hemopexin/hemopexin
A=tests/golden/adeno_fiber/adeno_fiber
run() { local t0=$(date +%s%N); "$@" 2>&1 | tr "\n" " "; local t1=$(date +%s%N); echo " WALL_ms=$(( (t1 - t0) / 1000000 ))"; }
echo "== hemopexin -dist_export"; for i in 1 2 3; do run famsa_amd/famsa-gpu -v -dist_export $H /tmp/o.csv; done
echo "== hemopexin -gt upgma"; for i in 1 2 3; do run famsa_amd/famsa-gpu -v -gt upgma -gt_export $H /tmp/o.dnd; done
echo "== hemopexin -gt sl"; for i in 1 2; do run famsa_amd/famsa-gpu -v -gt sl -gt_export $H /tmp/o.dnd; done
echo "== adeno -gt sl"; for i in 1 2; do run famsa_amd/famsa-gpu -v -gt sl -gt_export $A /tmp/o.dnd; done
echo "== hemopexin -gt upgma, LCSGPU_LANES=1"; LCSGPU_LANES=1 run famsa_amd/famsa-gpu -v -gt upgma -gt_export $H /tmp/o.dnd
cat > /tmp/t.cpp <<EOT
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main(){auto t0=std::chrono::steady_clock::now(); int n; hipGetDeviceCount(&n); auto t1=std::chrono::steady_clock::now(); hipSetDevice(0); hipFree(0); auto t2=std::chrono::steady_clock::now(); hipStream_t s; hipStreamCreate(&s); auto t3=std::chrono::steady_clock::now();
auto d=[](auto a,auto b){return std::chrono::duration<double>(b-a).count();}; printf("bare HIP: device count %.3f s, first context call %.3f s, stream %.3f s",d(t0,t1),d(t1,t2),d(t2,t3));}
EOT
/opt/rocm/bin/hipcc -O2 -o /tmp/t /tmp/t.cpp 2>/dev/null && { run /tmp/t; run /tmp/t; }
echo "== python reference timing (oracle/_ref, same box)"
python - <<'EOP'
import sys, time, os
sys.path.insert(0, "tests")
import oracle_bind
ref = oracle_bind.Ref()
H = "tests/golden/hemopexin/hemopexin"
for what in ("dist_export", "upgma"):
    for th in (16, 32):
        h = ref.open_fasta(H); t0 = time.time()
        if what == "dist_export": ref.dist_export(h, "/tmp/ref.csv", threads=th)
        else: ref.tree(h, "upgma", threads=th)
        print(f"reference {what} threads={th}: {time.time()-t0:.3f} s (library call only, no process start)"); ref.close(h)
EOP
