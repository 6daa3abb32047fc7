#!/bin/bash
# ASan+UBSan run of the C++ host layer (tree builders, FastTree with worker threads, writers) on
# oracle-supplied LCS matrices; CPU only.  Usage: scripts/asan_check.sh   (prints "asan ok" x3, then "tsan ok" x2)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); W=/tmp/asan; mkdir -p $W
cat > $W/main.cpp <<'CPP'
#include <cstdio>
#include <fstream>
#include <iterator>
#include <vector>
#include <cstdint>
extern "C" {
long famsa_host_tree_from_matrix(const char*, const uint32_t*, const char*, int, int, int, int, int, int, float, int, char*, long);
int famsa_host_dist_export_from_matrix(const char*, const uint32_t*, int, int, int, const char*);
const char* famsa_host_last_error(void);
}
int main(int, char** argv) {
    std::ifstream f(argv[2], std::ios::binary); std::vector<char> raw((std::istreambuf_iterator<char>(f)), {});
    const uint32_t* sq = (const uint32_t*)raw.data();
    std::vector<char> out(1 << 24);
    const char* gts[] = {"sl", "slink", "upgma", "nj", "upgma_modified"};
    for (const char* gt : gts) for (int heur = 0; heur < 3; ++heur) for (int keep = 0; keep < 2; ++keep)
        if (famsa_host_tree_from_matrix(argv[1], sq, gt, 1, keep, heur, 8, 40, 30, 0.3f, 2, out.data(), (long)out.size()) < 0) {
            printf("ERR %s %d %d: %s\n", gt, heur, keep, famsa_host_last_error()); return 1; }
    for (int s = 0; s < 2; ++s) for (int p = 0; p < 2; ++p)
        if (famsa_host_dist_export_from_matrix(argv[1], sq, 1, s, p, "/tmp/asan/o.csv")) { printf("ERR csv\n"); return 1; }
    printf("asan ok\n"); return 0;
}
CPP
cd $ROOT && python3 - <<'PY'
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, oracle_bind
from famsa_amd import seqio
o=oracle_bind.Oracle()
for name,f,cut in [('hemo','tests/golden/hemopexin/hemopexin',700),('adv','tests/golden/adversarial_tree.fasta',None),('dup','tests/golden/adeno_fiber_duplicates/adeno_fiber_duplicates',None)]:
    ids,seqs=seqio.read_fasta(f); ids,seqs=ids[:cut],seqs[:cut]; enc=[o.encode(s) for s in seqs]
    codes,off=seqio.pack(enc); n=len(enc)
    with open(f'/tmp/asan/{name}.fasta','w') as g:
        for i,s in zip(ids,seqs): g.write(i+'\n'+s+'\n')
    o.rect(codes,off,np.arange(n),np.arange(n)).astype(np.uint32).tofile(f'/tmp/asan/{name}.mat')
PY
cd $ROOT/famsa_amd/host && g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -pthread -o $W/check $W/main.cpp \
  seqset.cpp lcs_source.cpp trees.cpp fasttree.cpp pipeline.cpp capi.cpp -L.. -llcsgpu -lz -Wl,-rpath,$ROOT/famsa_amd
for n in hemo adv dup; do FAMSA_HOST_THREADS=4 ASAN_OPTIONS=detect_leaks=0 $W/check $W/$n.fasta $W/$n.mat; done
# ThreadSanitizer pass over the same driver (task pool of the FastTree recursion, parallel reader / sort)
g++ -std=c++17 -O1 -g -fsanitize=thread -fno-omit-frame-pointer -pthread -o $W/check_tsan $W/main.cpp \
  seqset.cpp lcs_source.cpp trees.cpp fasttree.cpp pipeline.cpp capi.cpp -L.. -llcsgpu -lz -Wl,-rpath,$ROOT/famsa_amd
for n in hemo dup; do FAMSA_HOST_THREADS=6 $W/check_tsan $W/$n.fasta $W/$n.mat 2>&1 | grep -E "WARNING: ThreadSanitizer|asan ok" | sed 's/asan ok/tsan ok/'; done
