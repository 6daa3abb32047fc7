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
#include <cstdlib>
extern "C" {
long famsa_host_tree_from_matrix(const char*, const uint32_t*, const char*, int, int, int, int, int, int, float, int, char*, long);
int famsa_host_dist_export_from_matrix(const char*, const uint32_t*, int, int, int, const char*);
const char* famsa_host_last_error(void);
int famsa_host_workset(const char*, int, int*, int*, int);
long famsa_host_newick(const int32_t*, const int32_t*, int, int, const char* const*, int, const int32_t*, char*, long);
}
#include <string>
// the parallel front end and the parallel Newick writer at a size that takes their many-thread paths: a FASTA of several MB
// (pieces of the reader, sorted runs + merges of the working order, duplicate flags) and a 100 000-leaf tree
static int big(const char* fasta, int n) {
    std::vector<int> a(n), b(n);
    for (int keep = 0; keep < 2; ++keep)
        if (famsa_host_workset(fasta, keep, a.data(), b.data(), n) < 0) { printf("ERR workset: %s\n", famsa_host_last_error()); return 1; }
    std::vector<int32_t> left(n - 1), right(n - 1), roots(n);
    for (int i = 0; i < n; ++i) roots[i] = i;
    unsigned x = 12345;
    for (int k = 0; k < n - 1; ++k) {
        x = x * 1664525u + 1013904223u; int i = x % roots.size(); std::swap(roots[i], roots.back()); int u = roots.back(); roots.pop_back();
        x = x * 1664525u + 1013904223u; int j = x % roots.size(); int v = roots[j];
        left[k] = u; right[k] = v; roots[j] = n + k;
    }
    std::vector<std::string> names(n); std::vector<const char*> np(n);
    for (int i = 0; i < n; ++i) { names[i] = ">s" + std::to_string(i); np[i] = names[i].c_str(); }
    std::vector<char> out((size_t)n * 24 + 64);
    if (famsa_host_newick(left.data(), right.data(), n, n - 1, np.data(), n, nullptr, out.data(), (long)out.size()) < 0) { printf("ERR newick\n"); return 1; }
    return 0;
}
int main(int argc, char** argv) {
    if (argc > 4 && big(argv[3], atoi(argv[4]))) return 1;
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
rng=np.random.Generator(np.random.PCG64(5)); A="ARNDCQEGHILKMFPSTWYV-"
with open('/tmp/asan/big.fasta','w') as g:
    seqs=[]
    for i in range(100000):
        s = seqs[int(rng.integers(0,len(seqs)))] if seqs and i%7==0 else "".join(A[c] for c in rng.integers(0,21,size=int(rng.integers(1,80))))
        seqs.append(s); g.write(">b%d\n%s\n"%(i,s))
PY
cd $ROOT/famsa_amd/host && g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -pthread -o $W/check $W/main.cpp \
  seqset.cpp lcs_source.cpp trees.cpp fasttree.cpp pipeline.cpp capi.cpp -L.. -llcsgpu -lz -Wl,-rpath,$ROOT/famsa_amd
for n in hemo adv dup; do FAMSA_HOST_THREADS=4 ASAN_OPTIONS=detect_leaks=0 $W/check $W/$n.fasta $W/$n.mat $W/big.fasta 100000; done
# ThreadSanitizer pass over the same driver (task pool of the FastTree recursion, parallel reader / sort)
g++ -std=c++17 -O1 -g -fsanitize=thread -fno-omit-frame-pointer -pthread -o $W/check_tsan $W/main.cpp \
  seqset.cpp lcs_source.cpp trees.cpp fasttree.cpp pipeline.cpp capi.cpp -L.. -llcsgpu -lz -Wl,-rpath,$ROOT/famsa_amd
for n in hemo dup; do FAMSA_HOST_THREADS=6 $W/check_tsan $W/$n.fasta $W/$n.mat $W/big.fasta 100000 2>&1 | grep -E "WARNING: ThreadSanitizer|asan ok" | sed 's/asan ok/tsan ok/'; done
