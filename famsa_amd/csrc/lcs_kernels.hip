// lcs_kernels.hip -- hand-written gfx950 (CDNA4) kernels for FAMSA's bit-parallel LCS.
//
// One lane = one PARTNER sequence (the streamed side), one workgroup = 4 waves = 256
// partners x R REF sequences (the bit-mask side).  The refs' occurrence masks
// M[ref][word][code] live in LDS; every lane keeps the bit-vector X of RG refs at a time
// in VGPRs (H x RG registers, H = number of 32-bit half-words = ceil(len_ref/32)) and walks
// its partner's residues.  Per 64-bit word-step the lane does one conflict-free ds_read_b64
// gather (address = residue code x 8 inside a 256-byte row, so the 20 residue codes hit 20
// distinct bank pairs) and six VALU ops, three per 32-bit half:
//     tB = V & M         v_and_b32         (a 2-source VOP2: 2.2 cycles; the 3-source form costs 2.6)
//     V2 = V + tB + c    v_add(c)_co_u32   -- one carry chain through all half-words
//     X  = V2 | (V & ~M) v_bitop3_b32      (three VGPR sources: 2.6 cycles if they sit in three different
//                                           register banks, 4.3 if not -- csrc/recolor_vgprs.py sees to that)
// which is the recurrence of CLCSBP_Classic_Impl (reference lcs/lcsbp_classic.h:51-58,
// 67-98) restated for 32-bit lanes.  Working in half-words saves the dead upper half of the
// last 64-bit word (400 aa: 13 half-words instead of 14).  No MFMA: this is integer/bit work,
// bounded by VALU issue, not by HBM.
//
// Exactness: the reference takes the carry out of a 64-bit word as (V2 < V) AFTER adding the
// carry-in (lcsbp_classic.h:55-56), which differs from a true 65-bit carry exactly when
// tB == ~0 and carry-in == 1 -- only possible if the ref has an aligned 64-residue
// homopolymer word at word index >= 1.  Such refs are flagged at upload and run through the
// QUIRK instantiations, which evaluate the reference's rule literally on 64-bit words; all
// other refs use the hardware carry chain, which is provably identical for them.
//
// Partner residues are stored position-major per 64-sequence tile ("column" layout):
// 16-byte chunk k of lane l of tile t sits at tile_base[t] + (k*64 + l)*16, so a wave's
// load of one chunk for 64 consecutive partners is one contiguous 1 KiB.  Each stored byte
// is (symbol code * 8) = the byte offset of that code inside an LDS mask row; padding is
// code 22 (UNKNOWN_SYMBOL, reference core/defs.h:66) whose mask is empty => a no-op step,
// exactly what the reference's `continue` (lcsbp_classic.h:82) / padded SIMD lanes do.
//
// Refs longer than 2048 residues (the reference's LoopCalculate case, lcsbp_classic.cpp:83)
// go through lcs_long_kernel: the ref is cut into segments of 16 words that are processed one
// after another with the same register-resident step; the carry that leaves the last word of
// a segment at partner position p is parked in a per-lane bit stream in global memory and
// re-enters word 0 of the next segment at the same position.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <stdint.h>

#include "lcs_kernels.h"

namespace lcsgpu {

typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((address_space(3))) uint64_t lds_u64;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

static constexpr uint32_t PAD4 = 0xB0B0B0B0u; // four bytes of 22*8

__device__ __forceinline__ uint32_t or_andn(uint32_t s, uint32_t v, uint32_t m)
{
    return __builtin_amdgcn_bitop3_b32(s, v, m, 0xF4); // s | (v & ~m): table index = 4 s + 2 v + m
}

// Mask read for the last, half-used word of an odd half-word count: only the low 32 bits are
// consumed.  ds_read_b32 banks are (addr/4)%32, so residue codes c and c+16 collide (2-way
// conflict on most of these reads; SQ_LDS_BANK_CONFLICT = 12% of LDS cycles), but a conflict-
// free ds_read_b64 of the full word measured 2-3% SLOWER end to end (n=40000: 533k vs 515k
// Gcell/s plain, 545k vs 540k pipelined) -- the extra LDS bytes cost more than the conflicts.
__device__ __forceinline__ uint64_t odd_half_read(const lds_u8* p)
{
#ifdef LCS_ODD_B64
    return *(const volatile lds_u64*)p;
#else
    return *(const lds_u32*)p;
#endif
}

// One partner residue against W 64-bit words of ONE ref with the reference's rule, literally
// (lcs/lcsbp_classic.h:51-58):  V2 = V + tB + cin;  cin' = (V2 < V).  Used for orientation-sensitive
// refs only (the QUIRK kernels); `row` = LDS address of the residue's entry in the ref's mask row 0.
// Returns the carry leaving the last word (needed by the long-ref kernel).
template <int W>
__device__ __forceinline__ unsigned literal_step(const lds_u8* row, uint32_t (&X)[2 * W], unsigned cin_bit)
{
    uint64_t cin = cin_bit;
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const uint64_t mm = *(const lds_u64*)(row + j * 256);
        const uint64_t V = ((uint64_t)X[2 * j + 1] << 32) | X[2 * j];
        const uint64_t tB = V & mm;
        const uint64_t V2 = V + tB + cin;
        cin = (V2 < V) ? 1u : 0u;
        const uint64_t Xn = V2 | (V & ~mm);
        X[2 * j] = (uint32_t)Xn;
        X[2 * j + 1] = (uint32_t)(Xn >> 32);
    }
    return (unsigned)cin;
}

// The maximum over the wave, as a value the compiler KNOWS to be wave-uniform (readfirstlane -> SGPR).  Without
// that, loops and branches on it are compiled as divergent control flow: exec-masked regions whose live-out
// vectors (the whole X array) must be kept for the "inactive" lanes in a second set of registers and copied at the
// join -- the kernels with a partial last chunk held 3 x H x RG extra VGPRs that way.
__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        int t = __shfl_xor(v, o, 64);
        v = t > v ? t : v;
    }
    return __builtin_amdgcn_readfirstlane(v);
}

// Occurrence masks of `count` 64-bit words of ref `rid`, starting at word `word0`, into LDS rows
// dst[w * 32 + c] = M[c][word0 + w]; M as CSequence::ComputeBitMasks builds it (reference core/sequence.cpp:190-201:
// bits only for codes < 20) -- built once per upload (upload_kernels.hip, masks_fill_kernel) and copied here.
// rid < 0 or a word beyond the ref fills "no match".  Called by all 256 threads of the workgroup.
__device__ __forceinline__ void build_mask_words(const RowsArgs& a, int rid, int word0, int count, lds_u64* dst,
                                                 int wave, int lane)
{
    const int tid = wave * 64 + lane;
    uint64_t base = 0;
    int have = 0; // words the ref holds from word0 on
    if (rid >= 0) {
        base = a.mask_base[rid];
        have = (int)(a.mask_base[rid + 1] - base) - word0;
    }
    for (int idx = tid; idx < count * 32; idx += 256) {
        const int w = idx >> 5;
        dst[idx] = w < have ? a.masks[(base + (uint64_t)(word0 + w)) * 32 + (idx & 31)] : 0ull;
    }
}

// The masks of a whole ref tile (nr_pad refs x W words) in one flat sweep: every thread copies its share of the
// nr_pad * W * 32 mask words, so the loads of different refs overlap instead of queueing ref after ref.
template <int W>
__device__ __forceinline__ void fill_tile_masks(const RowsArgs& a, int ref0, int nr, int nr_pad, lds_u64* dst, int tid)
{
    const int total = nr_pad * (W * 32);
    const int step = (int)blockDim.x; // 256, or 64 in the one-wave form (RowsArgs::block_threads)
#pragma unroll 4
    for (int idx = tid; idx < total; idx += step) {
        const int r = idx / (W * 32), rem = idx - r * (W * 32), w = rem >> 5;
        uint64_t m = 0;
        if (r < nr) {
            const int rid = a.ref_ids ? a.ref_ids[ref0 + r] : a.ref_begin + ref0 + r;
            const uint64_t base = a.mask_base[rid];
            if ((uint64_t)w < a.mask_base[rid + 1] - base) m = a.masks[(base + (uint64_t)w) * 32 + (rem & 31)];
        }
        dst[idx] = m;
    }
}

__device__ __forceinline__ void store_result(const RowsArgs& a, int k, int c, uint32_t res)
{
    const int64_t row = a.ref_rows ? a.ref_rows[k] : (int64_t)k + a.row0;
    int64_t idx;
    if (a.jobs && a.mode == MODE_RECT) { // batched rectangles: a row of its own per ref, columns relative to the job's first
        idx = a.ref_out0[k] + (c - a.ref_col0[k]);
    } else if (a.jobs) { // batched triangles: positions relative to the start of the ref's own id list
        if (c >= row)
            return;
        const int64_t g0 = a.ref_col0[k], lr = row - g0;
        idx = a.ref_out0[k] + lr * (lr - 1) / 2 + (c - g0);
    } else if (a.mode == MODE_TRIANGLE) { // rows/columns are POSITIONS in the caller's lists; only column < row
        if (c >= row)
            return;
        idx = row * (row - 1) / 2 + c - a.out_offset;
    } else {
        idx = row * a.ld + c;
    }
    if (a.elem_size == 2)
        ((uint16_t*)a.out)[idx] = (uint16_t)res;
    else
        ((uint32_t*)a.out)[idx] = res;
}

// bytes of mask rows in LDS ahead of the fused launch's records
__device__ __forceinline__ lds_u64* fuse_lds(unsigned char* smem, int refs_per_block, int W)
{
    return (lds_u64*)(smem + (size_t)refs_per_block * W * 256);
}

// ---- the local half of a Boruvka round fused into the LCS launch (FuseArgs, lcs_kernels.h) ------------------------
typedef unsigned long long fuse_rec;
static constexpr fuse_rec FUSE_NONE = ~0ull;
__device__ __forceinline__ fuse_rec fuse_pack(uint32_t l, uint32_t len, uint32_t u)
{
    return ((fuse_rec)l << 48) | ((fuse_rec)(len & 0xFFFFu) << 32) | u;
}
__device__ __forceinline__ uint32_t fuse_l(fuse_rec r) { return (uint32_t)(r >> 48); }
__device__ __forceinline__ uint32_t fuse_len(fuse_rec r) { return (uint32_t)(r >> 32) & 0xFFFFu; }
__device__ __forceinline__ uint32_t fuse_u(fuse_rec r) { return (uint32_t)r; }

// Transform<double, kind> (reference tree/AbstractTreeGenerator.hpp:28-82): host-built pow table + IEEE division
__device__ __forceinline__ double fuse_dist(const FuseArgs& f, uint32_t l, uint32_t indel)
{
    if (l == 0) return 1.7976931348623155e308; // nextafter(DBL_MAX, 0), hpp:61,73
    return f.kind == 1 ? f.pow_table[indel] / (double)l : (double)indel / (double)l;
}

// Is record A a better edge for the vertex v (length len_v) than record B?  MSTPrim's order (tree/MSTPrim.cpp:493-509):
// smaller distance, then the LARGER packed id pair -- for a fixed v and one side (all u < v, or all u > v) that is the
// larger u.  Equal (l, length) give the same distance bit for bit: no division then.
__device__ __forceinline__ bool fuse_better(const FuseArgs& f, fuse_rec A, fuse_rec B, uint32_t len_v)
{
    if (A == FUSE_NONE) return false;
    if (B == FUSE_NONE) return true;
    const uint32_t la = fuse_l(A), lb = fuse_l(B), na = fuse_len(A), nb = fuse_len(B);
    if (la == lb && na == nb) return fuse_u(A) > fuse_u(B);
    const double da = fuse_dist(f, la, len_v + na - 2u * la), db = fuse_dist(f, lb, len_v + nb - 2u * lb);
    if (da != db) return da < db;
    return fuse_u(A) > fuse_u(B);
}

// The integer pre-filter (mst_kernels.hip, l_threshold): with the edge `rec` in hand, a candidate whose other endpoint
// is len_c long can reach its distance only with l >= this.
__device__ __forceinline__ uint32_t fuse_thr(fuse_rec rec, uint32_t len_c)
{
    if (rec == FUSE_NONE) return 0;
    const uint32_t l_b = fuse_l(rec), len_b = fuse_len(rec);
    const uint32_t slack = len_b > len_c ? (len_b - len_c + 1) >> 1 : 0;
    return l_b > slack ? l_b - slack : 0;
}

__device__ __forceinline__ void fuse_global_update(const FuseArgs& f, fuse_rec* slot, fuse_rec mine, uint32_t len_v)
{
    if (mine == FUSE_NONE) return;
    fuse_rec cur = __atomic_load_n(slot, __ATOMIC_RELAXED);
    while (cur != mine && fuse_better(f, mine, cur, len_v)) {
        const fuse_rec prev = atomicCAS(slot, cur, mine);
        if (prev == cur) break;
        cur = prev;
    }
}

// LDS of a fused launch, behind the masks: f_col[256] -- the best edge of this workgroup's column vertices, each
// owned by its lane -- and f_row[4][32] -- the best edge of its row vertices as seen by each wave.  Both start
// from the global records (hints: whatever other workgroups have folded so far) and are folded back at the end.
__device__ __forceinline__ void fuse_init(const RowsArgs& a, lds_u64* f_col, lds_u64* f_row, int ref0, int nr, int c, bool valid,
                                          int tid)
{
    f_col[tid] = valid ? a.fuse.col_rec[a.col_begin + c] : FUSE_NONE;
    if (tid < 128) {
        const int i = tid & 31;
        fuse_rec h = FUSE_NONE;
        if (i < nr) h = a.fuse.row_rec[a.ref_ids ? a.ref_ids[ref0 + i] : a.ref_begin + ref0 + i];
        f_row[tid] = h;
    }
}

// NR results of this lane (refs kl0 .. kl0 + cnt - 1 of the workgroup's tile against column c).  The bulk costs two
// integer thresholds and a compare per result; the exact comparison (component labels, the f64 division, a wave
// reduction for the row side) runs only when some lane's LCS length reaches a threshold.
template <int NR>
__device__ __forceinline__ void fuse_group(const RowsArgs& a, lds_u64* f_col, lds_u64* f_row, int ref0, int kl0, int cnt, int c_in,
                                           int col_limit, const uint32_t (&res)[NR], int tid_in)
{
    // Everything here that does not depend on the ref group -- the column's id, its length, its LDS slot -- would be
    // hoisted out of the workgroup's loop over ref groups by the compiler and then stay in VGPRs through the hot loop
    // (+17 registers: 5 -> 4 waves per SIMD at 13 half-words).  The two lane indices are made opaque here, so all of
    // it is recomputed per group, after the loop: a dozen instructions against ~60 000.
    int c = c_in, tid = tid_in;
    asm volatile("" : "+v"(c), "+v"(tid));
    const FuseArgs& f = a.fuse;
    const bool valid = c < col_limit;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pid = a.col_begin + c;
    const uint32_t len_c = valid ? a.lens[pid] : 0u;
    fuse_rec crec = f_col[tid];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        if (r >= cnt) break;
        const int rid = a.ref_ids ? a.ref_ids[ref0 + kl0 + r] : a.ref_begin + ref0 + kl0 + r;
        const uint32_t len_r = a.lens[rid];
        const fuse_rec rrec = f_row[wave * 32 + kl0 + r];
        const uint32_t l = res[r];
        const bool pass = valid && pid < rid && (l >= fuse_thr(rrec, len_c) || l >= fuse_thr(crec, len_r));
        if (__builtin_amdgcn_ballot_w64(pass) == 0) continue;
        bool cand = pass;
        if (f.comp && cand) cand = f.comp[pid] != f.comp[rid];
        if (__builtin_amdgcn_ballot_w64(cand) == 0) continue;
        fuse_rec mine = 0;
        unsigned long long dbits = 0;
        if (cand) {
            const fuse_rec as_col = fuse_pack(l, len_r, (uint32_t)rid); // the column vertex's view of the edge
            if (fuse_better(f, as_col, crec, len_c)) crec = as_col;
            mine = fuse_pack(l, len_c, (uint32_t)pid);                  // the row vertex's view
            dbits = (unsigned long long)__double_as_longlong(fuse_dist(f, l, len_r + len_c - 2u * l));
        }
        // the wave's best candidate for the row vertex -- smaller d (distances are >= 0: their bits order like the
        // values), then larger u -- by a walk over the candidate lanes in SGPRs (few lanes get here; a shuffle
        // butterfly would keep six lane-index registers alive through the hot loop)
        unsigned long long todo = __builtin_amdgcn_ballot_w64(cand);
        unsigned long long bd = 0;
        fuse_rec bm = FUSE_NONE;
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const unsigned long long d2 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(dbits >> 32), src) << 32) |
                                          (uint32_t)__builtin_amdgcn_readlane((int)dbits, src);
            const fuse_rec m2 = ((fuse_rec)(uint32_t)__builtin_amdgcn_readlane((int)(mine >> 32), src) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)mine, src);
            if (bm == FUSE_NONE || d2 < bd || (d2 == bd && fuse_u(m2) > fuse_u(bm))) {
                bd = d2;
                bm = m2;
            }
        }
        mine = bm;
        if ((tid & 63) == 0 && fuse_better(f, mine, rrec, len_r)) f_row[wave * 32 + kl0 + r] = mine;
    }
    f_col[tid] = crec;
}

// The length bound for a whole tile (FuseArgs::prune; call after fuse_init and a barrier): true if no pair of the tile can
// improve a record.  A pair's distance is at least Transform(lcs = lmax, indel = gap) with gap = the distance between the
// length ranges of the tile's rows and columns and lmax = the longest LCS their lengths allow: the pow table and IEEE
// division are monotone, so the bound never exceeds a pair's distance; the comparison is STRICT, so an equal distance --
// which the id order might prefer -- is never let go.  Records only improve: a stale (larger) one only makes the test weaker.
__device__ __forceinline__ bool fuse_tile_is_useless(const RowsArgs& a, const lds_u64* f_col, const lds_u64* f_row, uint32_t* red,
                                                     int ref0, int nr, int c, int col_limit, int tid)
{
    const FuseArgs& f = a.fuse;
    int rid_last = 0; // the tile's largest row (fused launches: row == ref id; the refs of a class need not be contiguous)
    for (int r = 0; r < nr; ++r) {
        const int rid = a.ref_ids ? a.ref_ids[ref0 + r] : a.ref_begin + ref0 + r;
        rid_last = rid > rid_last ? rid : rid_last;
    }
    const int pid = a.col_begin + c;
    const bool has = c < col_limit && pid < rid_last; // this column has a pair in the tile
    uint32_t lmin = 0xffffffffu, lmax = 0u, open = 0u; // open: a vertex without a record (nothing bounds it)
    unsigned long long dmax = 0ull;                    // distances are >= 0: their bits order like the values
    auto take = [&](fuse_rec rec, uint32_t len_v) {
        lmin = len_v < lmin ? len_v : lmin;
        lmax = len_v > lmax ? len_v : lmax;
        if (rec == FUSE_NONE) {
            open = 1u;
            return;
        }
        const uint32_t l = fuse_l(rec);
        const unsigned long long d = (unsigned long long)__double_as_longlong(fuse_dist(f, l, len_v + fuse_len(rec) - 2u * l));
        dmax = d > dmax ? d : dmax;
    };
    if (has) take(f_col[tid], a.lens[pid]);
    // the columns' ranges: wave by wave, then through LDS
    uint32_t cmin = lmin, cmax = lmax, copen = open;
    unsigned long long cd = dmax;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t m2 = __shfl_xor(cmin, o, 64), x2 = __shfl_xor(cmax, o, 64), o2 = __shfl_xor(copen, o, 64);
        const unsigned long long d2 = __shfl_xor(cd, o, 64);
        cmin = m2 < cmin ? m2 : cmin;
        cmax = x2 > cmax ? x2 : cmax;
        copen |= o2;
        cd = d2 > cd ? d2 : cd;
    }
    const int wave = tid >> 6;
    if ((tid & 63) == 0) {
        red[wave * 8 + 0] = cmin;
        red[wave * 8 + 1] = cmax;
        red[wave * 8 + 2] = copen;
        red[wave * 8 + 3] = (uint32_t)cd;
        red[wave * 8 + 4] = (uint32_t)(cd >> 32);
    }
    // the rows (at most 32: the first wave's lanes), every wave's view of a row folded
    lmin = 0xffffffffu; lmax = 0u; open = 0u; dmax = 0ull;
    if (tid < nr) {
        const int rid = a.ref_ids ? a.ref_ids[ref0 + tid] : a.ref_begin + ref0 + tid;
        const uint32_t len_r = a.lens[rid];
        fuse_rec best = f_row[tid]; // (all four waves' copies start equal: fuse_init)
        take(best, len_r);
    }
    if (wave == 0) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t m2 = __shfl_xor(lmin, o, 64), x2 = __shfl_xor(lmax, o, 64), o2 = __shfl_xor(open, o, 64);
            const unsigned long long d2 = __shfl_xor(dmax, o, 64);
            lmin = m2 < lmin ? m2 : lmin;
            lmax = x2 > lmax ? x2 : lmax;
            open |= o2;
            dmax = d2 > dmax ? d2 : dmax;
        }
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t kmin = 0xffffffffu, kmax = 0u, kopen = 0u;
        unsigned long long kd = 0ull;
        for (int w = 0; w < 4; ++w) {
            kmin = red[w * 8 + 0] < kmin ? red[w * 8 + 0] : kmin;
            kmax = red[w * 8 + 1] > kmax ? red[w * 8 + 1] : kmax;
            kopen |= red[w * 8 + 2];
            const unsigned long long d = ((unsigned long long)red[w * 8 + 4] << 32) | red[w * 8 + 3];
            kd = d > kd ? d : kd;
        }
        bool useless = false;
        if (kmax == 0u) useless = true; // no pair at all in this tile
        else if (!kopen && !open && lmax != 0u) {
            // rows [lmin, lmax], columns [kmin, kmax]
            const uint32_t gap = kmin > lmax ? kmin - lmax : (lmin > kmax ? lmin - kmax : 0u);
            const uint32_t l_most = lmax < kmax ? lmax : kmax;
            if (gap > 0u && l_most > 0u) {
                const unsigned long long bound = (unsigned long long)__double_as_longlong(fuse_dist(f, l_most, gap));
                const unsigned long long worst = kd > dmax ? kd : dmax;
                useless = bound > worst;
            }
        }
        red[7] = useless ? 1u : 0u;
        if (f.stats) atomicAdd(f.stats + (useless ? 1 : 0), 1ull);
    }
    __syncthreads();
    return red[7] != 0u;
}

// after the workgroup's last result (and a barrier): fold the records back into the global ones
__device__ __forceinline__ void fuse_flush(const RowsArgs& a, const lds_u64* f_col, const lds_u64* f_row, int ref0, int nr, int c,
                                           bool valid, int tid)
{
    const FuseArgs& f = a.fuse;
    if (tid < nr) {
        const int rid = a.ref_ids ? a.ref_ids[ref0 + tid] : a.ref_begin + ref0 + tid;
        const uint32_t len_r = a.lens[rid];
        fuse_rec best = f_row[tid];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const fuse_rec o = f_row[w * 32 + tid];
            if (o != best && fuse_better(f, o, best, len_r)) best = o;
        }
        fuse_global_update(f, &f.row_rec[rid], best, len_r);
    }
    if (valid) {
        const int pid = a.col_begin + c;
        fuse_global_update(f, &f.col_rec[pid], f_col[tid], a.lens[pid]);
    }
}

// Workgroup -> (column block x, ref tile y).  In the compact triangle grid the rows are walked
// from the bottom (most column blocks) to the top, so the chip fills at once and the workgroups
// above the diagonal -- half of a 2-D grid, and nearly all of its first rows -- are never launched
// (measured at n = 10 000: the dispatcher spent ~3 ms of an 18 ms launch retiring them).
__device__ __forceinline__ void block_coords(const RowsArgs& a, int& x, int& y)
{
    if (!a.tri_prefix) {
        // 2-D grid: workgroups go to the 8 XCDs round robin by their linear id.  With the column block as the fast
        // index and a column count that is a multiple of 8, an XCD would own whole COLUMNS -- and in triangle mode the
        // first columns hold all the work (4000 long sequences, 16 column blocks: XCD 0 had 744 workgroups below the
        // diagonal, XCD 7 had 296; the long-ref kernel ran with half of the CUs idle).  So the ref tile is the fast index.
        const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
        y = (int)(lin % gridDim.y);
        x = (int)(lin / gridDim.y);
        return;
    }
    const int bid = blockIdx.x;
    if (a.diag_prefix) { // by diagonals: the d-th tile from the row's end, rows in the compact grid's order (fullest first)
        int lo = 0, hi = a.diag_count - 1; // largest d with diag_prefix[d] <= bid
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (a.diag_prefix[mid] <= bid) lo = mid; else hi = mid - 1;
        }
        const int k = bid - a.diag_prefix[lo]; // rows with more than `lo` column blocks are a prefix of the order
        y = a.tri_rows - 1 - k;
        x = a.tri_prefix[k + 1] - a.tri_prefix[k] - 1 - lo;
        return;
    }
    int lo = 0, hi = a.tri_rows - 1; // largest k with tri_prefix[k] <= bid
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.tri_prefix[mid] <= bid) lo = mid; else hi = mid - 1;
    }
    y = a.tri_rows - 1 - lo;
    x = bid - a.tri_prefix[lo];
}

// The tile of this workgroup: refs [ref0, ref0 + nr) of the launch's ref list against the partner
// positions [c0, c0 + 256) below `col_limit`.
__device__ __forceinline__ void block_tile(const RowsArgs& a, int R, int& ref0, int& nr, int& c0, int& col_limit)
{
    if (a.jobs) {
        const int4 j = a.jobs[blockIdx.x];
        ref0 = j.x;
        nr = j.y;
        c0 = j.z;
        col_limit = j.w;
        return;
    }
    int bx, by;
    block_coords(a, bx, by);
    ref0 = by * R;
    nr = min(R, a.n_refs - ref0);
    c0 = bx * 256;
    col_limit = a.n_cols;
}

// Triangle mode: a block whose columns all lie at or beyond its largest row has no work.
__device__ __forceinline__ bool block_is_above_diagonal(const RowsArgs& a, int ref0, int nr, int c0)
{
    if (a.mode != MODE_TRIANGLE)
        return false;
    int64_t max_row = 0;
    for (int r = 0; r < nr; ++r) {
        const int64_t row = a.ref_rows ? a.ref_rows[ref0 + r] : (int64_t)(ref0 + r) + a.row0;
        max_row = row > max_row ? row : max_row;
    }
    return c0 >= max_row;
}

// Orientation-sensitive refs (SURVEY note Q): one ref at a time, whole 64-bit words, literal rule.
// Rare by construction (a ref needs an aligned 64-residue homopolymer), so this kernel is written
// for exactness, not speed.
template <int H, bool FUSE>
__global__ __launch_bounds__(256) void lcs_rows_kernel_quirk(RowsArgs a)
{
    static_assert(H % 2 == 0, "quirk instantiations use whole 64-bit words");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int W = H / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int R = a.refs_per_block;
    int ref0, nr, c0, col_limit;
    block_tile(a, R, ref0, nr, c0, col_limit);
    if (block_is_above_diagonal(a, ref0, nr, c0))
        return;
    for (int r = 0; r < nr; ++r) {
        const int rid = a.ref_ids ? a.ref_ids[ref0 + r] : a.ref_begin + ref0 + r;
        build_mask_words(a, rid, 0, W, (lds_u64*)smem + r * W * 32, wave, lane);
    }
    const int c = c0 + tid;
    const bool valid = c < col_limit;
    [[maybe_unused]] lds_u64* f_col = fuse_lds(smem, R, W);
    [[maybe_unused]] lds_u64* f_row = f_col + 256;
    if constexpr (FUSE) fuse_init(a, f_col, f_row, ref0, nr, c, valid, tid);
    __syncthreads();

    const int pid = a.col_ids ? a.col_ids[valid ? c : 0] : a.col_begin + (valid ? c : 0);
    const uint32_t len_p = valid ? a.lens[pid] : 0u;
    const uint8_t* pbase = a.tiles + a.tile_base[pid >> 6] + (uint64_t)(pid & 63) * 16;
    const int my_chunks = (int)((len_p + 15) >> 4);
    const int wave_chunks = wave_max(my_chunks);

    for (int r = 0; r < nr; ++r) {
        uint32_t X[H];
#pragma unroll
        for (int j = 0; j < H; ++j)
            X[j] = ~0u;
        const lds_u8* masks = (const lds_u8*)smem + r * (W * 256);
        for (int k = 0; k < wave_chunks; ++k) {
            uint4 q = make_uint4(PAD4, PAD4, PAD4, PAD4);
            if (k < my_chunks)
                q = *(const uint4*)(pbase + (size_t)k * 1024);
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int b = 0; b < 16; ++b)
                literal_step<W>(masks + ((w[b >> 2] >> (8 * (b & 3))) & 0xFFu), X, 0u);
        }
        uint32_t res[1] = {0}; // number of zero bits (reference lcsbp_classic.h:60-65)
#pragma unroll
        for (int j = 0; j < H; ++j)
            res[0] += __popc(~X[j]);
        if (valid && (!FUSE || a.out)) store_result(a, ref0 + r, c, res[0]);
        if constexpr (FUSE) fuse_group<1>(a, f_col, f_row, ref0, r, 1, c, col_limit, res, tid);
    }
    if constexpr (FUSE) {
        __syncthreads();
        fuse_flush(a, f_col, f_row, ref0, nr, c, valid, tid);
    }
}

// ---- software-pipelined inner loop -------------------------------------------------------
// The compiler's own schedule issues the mask gathers in bursts right before their first use
// and pays the LDS latency on every burst, and it orders the two halves of a word so that an
// s_nop is needed between the dependent v_addc pair.  Here the instruction stream of one
// 16-residue chunk (16 x RG x W word-steps) is laid out explicitly and pinned with
// sched_barrier: the gather for word-step t+LOOKAHEAD is issued before the VALU work of
// word-step t (ring of LOOKAHEAD mask registers, carried across the chunk loop), and each word
// runs  tB.lo, add.lo, X.lo, tB.hi, addc.hi, X.hi  so two instructions always sit between a
// carry producer and its consumer (the gfx950 VALU-writes-VCC -> VALU-reads-VCC distance).
#define LCS_PIN() __builtin_amdgcn_sched_barrier(0)
#ifndef LCS_LOOKAHEAD
#define LCS_LOOKAHEAD 4 // measured at n=40000: 4 / 8 / 16 -> 553.8 / 550.1 / 550.2 Tcell/s
#endif

template <int H, int RG, int LOOKAHEAD>
struct Pipe {
    static constexpr int W = (H + 1) / 2;
    static constexpr int PER_POS = RG * W;     // word-steps per residue
    static constexpr int T = 16 * PER_POS;     // word-steps per chunk
    static_assert(T % LOOKAHEAD == 0, "the ring phase must be the same at every chunk boundary");

    static __device__ __forceinline__ uint32_t byte_of(const uint4& q, int b)
    {
        const uint32_t w = (b >> 2) == 0 ? q.x : (b >> 2) == 1 ? q.y : (b >> 2) == 2 ? q.z : q.w;
        return (w >> (8 * (b & 3))) & 0xFFu;
    }

    // mask word of word-step t2 (t2 may run into the next chunk)
    static __device__ __forceinline__ uint64_t gather(const lds_u8* grp, const uint4& q, const uint4& qn, int t2)
    {
        const int b2 = t2 / PER_POS, rem = t2 - b2 * PER_POS;
        const uint32_t code8 = b2 < 16 ? byte_of(q, b2) : byte_of(qn, b2 - 16);
        if ((H & 1) && (rem % W) == W - 1)
            return odd_half_read(grp + code8 + rem * 256);
        return *(const lds_u64*)(grp + code8 + rem * 256);
    }

    static __device__ __forceinline__ void prime(const lds_u8* grp, const uint4& q, uint64_t (&ring)[LOOKAHEAD])
    {
#pragma unroll
        for (int t = 0; t < LOOKAHEAD; ++t)
            ring[t] = gather(grp, q, q, t);
    }

    // CARRY (long refs, one segment of the bit-vector per pass): bit b of `cw` enters word 0 at
    // residue b, the carry leaving the last word is collected in bit b of `cout`.
    // NPOS < 16: the partner's LAST chunk, of which only the first NPOS residues can be real for any lane of
    // the wave (the rest is padding, a no-op step each): the tail of a 100-residue partner costs 4 positions
    // instead of 16 (-11% of its steps).
    // AHEAD (with NPOS < 16): the body is one turn of a loop over groups of NPOS residues of the same chunk -- the
    // gathers run ahead into the residues that follow (the caller shifts q by NPOS bytes after every turn).
    template <bool CARRY = false, int NPOS = 16, bool AHEAD = false>
    static __device__ __forceinline__ void chunk(const lds_u8* grp, const uint4& q, const uint4& qn,
                                                 uint64_t (&ring)[LOOKAHEAD], uint32_t (&X)[RG][H], uint32_t cw = 0,
                                                 uint32_t* cout_p = nullptr)
    {
        static_assert(!CARRY || RG == 1, "carry streams are per (ref, partner)");
        static_assert(NPOS == 16 || !CARRY, "partial chunks exist in the register-resident kernel only");
        static_assert((NPOS * PER_POS) % LOOKAHEAD == 0, "ring phase");
        uint32_t cout = 0;
#pragma unroll
        for (int b = 0; b < NPOS; ++b) {
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                unsigned cin = 0;
                if (CARRY) {
                    cin = cw & 1u;
                    cw >>= 1;
                    LCS_PIN();
                }
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    const int t = (b * RG + r) * W + j;
                    const uint64_t mm = ring[t % LOOKAHEAD];
                    if (NPOS == 16 || AHEAD || t + LOOKAHEAD < NPOS * PER_POS) // nothing follows a partial chunk
                        ring[t % LOOKAHEAD] = gather(grp, q, qn, t + LOOKAHEAD);
                    LCS_PIN();
                    const uint32_t m0 = (uint32_t)mm, m1 = (uint32_t)(mm >> 32);
                    unsigned co;
                    {
                        const uint32_t V = X[r][2 * j];
                        const uint32_t tb = V & m0;
                        LCS_PIN();
                        const uint32_t s = __builtin_addc(V, tb, cin, &co);
                        LCS_PIN();
                        X[r][2 * j] = or_andn(s, V, m0);
                        cin = co;
                        LCS_PIN();
                    }
                    if (2 * j + 1 < H) {
                        const uint32_t V = X[r][2 * j + 1];
                        const uint32_t tb = V & m1;
                        LCS_PIN();
                        const uint32_t s = __builtin_addc(V, tb, cin, &co);
                        LCS_PIN();
                        X[r][2 * j + 1] = or_andn(s, V, m1);
                        cin = co;
                        LCS_PIN();
                    }
                }
                if (CARRY) {
                    cout = (cout >> 1) | (cin << 15);
                    asm volatile("" : "+v"(cout)); // materialise now: otherwise 16 carries wait in SGPR pairs
                    LCS_PIN();
                }
            }
        }
        if (CARRY) *cout_p = cout;
    }
};

// (The instantiations with X = 64 registers -- 16 x 4, 32 x 2, 64 x 1 half-words x refs -- hold 105 VGPRs = 4 waves per
// SIMD; asked for 5 through the launch bounds the compiler fits 96 with three dwords spilled outside the loop bodies,
// and nothing is gained: 448 / 512 / 1024 / 2048 aa 615 / 616 / 615 / 611 Tcell/s with 4 waves, 619 / 611 / 613 / 606 with 5.)
// FUSE: the instantiations that also fold their results into the per-vertex best-edge records (FuseArgs); a separate
// set of kernels (built as its own translation unit, LCS_FUSED_TU) so that the plain ones keep their registers: the
// fold costs the 13-half-word kernel 5 VGPRs (90 -> 95, still 5 waves per SIMD), the 14-half-word one its fifth wave.
template <int H, int RG, int LOOKAHEAD, bool FUSE>
__global__ __launch_bounds__(256) void lcs_rows_kernel_pipe(RowsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int W = (H + 1) / 2;
    using P = Pipe<H, RG, LOOKAHEAD>;
    const int tid = threadIdx.x;
    const int R = a.refs_per_block;
    int ref0, nr, c0, col_limit;
    block_tile(a, R, ref0, nr, c0, col_limit);
    if (block_is_above_diagonal(a, ref0, nr, c0))
        return;
    fill_tile_masks<W>(a, ref0, nr, (nr + RG - 1) / RG * RG, (lds_u64*)smem, tid);
    const int c = c0 + tid;
    const bool valid = c < col_limit;
    [[maybe_unused]] lds_u64* f_col = fuse_lds(smem, R, W);
    [[maybe_unused]] lds_u64* f_row = f_col + 256;
    if constexpr (FUSE) fuse_init(a, f_col, f_row, ref0, nr, c, valid, tid);
    __syncthreads();
    if constexpr (FUSE) {
        if (a.fuse.prune && fuse_tile_is_useless(a, f_col, f_row, (uint32_t*)(f_row + 4 * 32), ref0, nr, c, col_limit, tid)) return;
    }

    const int pid = a.col_ids ? a.col_ids[valid ? c : 0] : a.col_begin + (valid ? c : 0);
    const uint32_t len_p = valid ? a.lens[pid] : 0u;
    const uint8_t* pbase = a.tiles + a.tile_base[pid >> 6] + (uint64_t)(pid & 63) * 16;
    const int my_chunks = (int)((len_p + 15) >> 4);
    const int wave_chunks = wave_max(my_chunks);
    // residues of the wave's last chunk that are real for some lane, in quads of 4
    const int my_tail = my_chunks == wave_chunks ? (int)(((len_p - 1) & 15) >> 2) + 1 : 0;
    const int tail_quads = wave_max(my_tail);

    for (int g = 0; g < nr; g += RG) {
        uint32_t X[RG][H];
#pragma unroll
        for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int j = 0; j < H; ++j)
                X[r][j] = ~0u;
        const lds_u8* grp = (const lds_u8*)smem + g * (W * 256);
        uint4 q = make_uint4(PAD4, PAD4, PAD4, PAD4);
        if (0 < my_chunks)
            q = *(const uint4*)pbase;
        uint64_t ring[LOOKAHEAD];
        P::prime(grp, q, ring);
        // The last chunk of the wave's longest partner runs as a loop over 4-residue quads, as many as that partner
        // needs: a 100-residue partner pays 100 steps, not 112.  One extra body of 4 residues per kernel; X stays in
        // the registers of the main loop.  (The first form -- three more copies of the body for 4 / 8 / 12 residues
        // behind a branch -- made the compiler keep a second set of X registers for them and copy at the join: 121
        // VGPRs for 8 half-words, 171 for 13, which is why only H <= 8 had it.)
        const int full_chunks = wave_chunks - 1;
        for (int k = 0; k < full_chunks; ++k) {
            uint4 qn = make_uint4(PAD4, PAD4, PAD4, PAD4);
            if (k + 1 < my_chunks)
                qn = *(const uint4*)(pbase + (size_t)(k + 1) * 1024);
            P::chunk(grp, q, qn, ring, X);
            q = qn;
        }
        for (int i = 0; i < tail_quads; ++i) {
            P::template chunk<false, 4, true>(grp, q, q, ring, X);
            q.x = q.y;
            q.y = q.z;
            q.z = q.w;
            q.w = PAD4;
        }
        uint32_t res[RG];
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            res[r] = 0;
#pragma unroll
            for (int j = 0; j < H; ++j)
                res[r] += __popc(~X[r][j]);
        }
        if (!FUSE || a.out) {
#pragma unroll
            for (int r = 0; r < RG; ++r)
                if (g + r < nr && valid) store_result(a, ref0 + g + r, c, res[r]);
        }
        if constexpr (FUSE) fuse_group<RG>(a, f_col, f_row, ref0, g, min(RG, nr - g), c, col_limit, res, tid);
    }
    if constexpr (FUSE) {
        __syncthreads();
        fuse_flush(a, f_col, f_row, ref0, nr, c, valid, tid);
    }
}

// ---- refs longer than 2048 residues -------------------------------------------------------
// The ref is cut into segments of at most SEGW words (X of one segment in 2 x SEGW VGPRs) that are
// processed one after another; the last segment of a ref uses the narrowest of the 8 / 16-word
// passes that covers what is left, so a 3000-residue ref (47 words) costs 16 + 16 + 16 words, a 2100-residue
// one (33 words) 16 + 16 + 8.  Carries between segments go through a per-lane stream of 16-bit words (one
// per 16-residue chunk) in global scratch: carry[(slot*n_chunks + k)*256 + tid], read and
// rewritten in place by each segment.
// Segment width, measured with the single-block chunk loop below (6000 sequences of 2100 / 3000 / 4200 / 6000
// residues, Tcell/s): 16 words (88 VGPRs, 5 waves per SIMD) 363 / 433 / 387 / 417, 24 words (108 VGPRs, 4 waves)
// 364 / 427 / 384 / 412, 32 words (124 VGPRs) 361 / 428 / 382 / 415 -- the narrowest is kept.
#ifndef LCS_SEGW
#define LCS_SEGW 16
#endif
static constexpr int SEGW = LCS_SEGW;

// one segment of W words of ref `rid`, starting at word `word0`, against this lane's partner
template <bool QUIRK, int W>
__device__ __forceinline__ uint32_t long_segment_pass(const RowsArgs& a, int rid, int word0, bool first, bool last,
                                                      unsigned char* smem, const uint8_t* pbase, int my_chunks,
                                                      int wave_chunks, uint16_t* my_carry, int wave, int lane)
{
    __syncthreads(); // previous segment's readers are done with the masks
    build_mask_words(a, rid, word0, W, (lds_u64*)smem, wave, lane);
    __syncthreads();
    uint32_t res = 0;
    if constexpr (!QUIRK) {
        // the pipelined half-word step of the hot kernel
        using P = Pipe<2 * W, 1, LCS_LOOKAHEAD>;
        uint32_t X[1][2 * W];
#pragma unroll
        for (int j = 0; j < 2 * W; ++j)
            X[0][j] = ~0u;
        uint4 q = make_uint4(PAD4, PAD4, PAD4, PAD4);
        if (0 < my_chunks)
            q = *(const uint4*)pbase;
        uint64_t ring[LCS_LOOKAHEAD];
        P::prime((const lds_u8*)smem, q, ring);
        // The chunk body must stay ONE basic block up to the loop latch: with a conditional carry store behind it
        // the compiler sank the last residue's 2W X updates past the branch and kept their 3 x 2W operands alive
        // over the whole body (24 words: 187 VGPRs, 2 waves per SIMD).  So the carry word is stored always (the
        // last segment's is never read) and the next chunk's incoming word is requested a chunk ahead, like qn.
        uint32_t cw = (!first && 0 < wave_chunks) ? my_carry[0] : 0u;
        for (int k = 0; k < wave_chunks; ++k) {
            uint4 qn = make_uint4(PAD4, PAD4, PAD4, PAD4);
            if (k + 1 < my_chunks)
                qn = *(const uint4*)(pbase + (size_t)(k + 1) * 1024);
            uint32_t cwn = 0;
            if (!first && k + 1 < wave_chunks)
                cwn = my_carry[(size_t)(k + 1) * 256];
            uint32_t cout = 0;
            P::template chunk<true>((const lds_u8*)smem, q, qn, ring, X, cw, &cout);
            my_carry[(size_t)k * 256] = (uint16_t)cout;
            q = qn;
            cw = cwn;
        }
#pragma unroll
        for (int j = 0; j < 2 * W; ++j)
            res += __popc(~X[0][j]);
    } else {
        uint32_t X[2 * W];
#pragma unroll
        for (int j = 0; j < 2 * W; ++j)
            X[j] = ~0u;
        for (int k = 0; k < wave_chunks; ++k) {
            uint4 q = make_uint4(PAD4, PAD4, PAD4, PAD4);
            if (k < my_chunks)
                q = *(const uint4*)(pbase + (size_t)k * 1024);
            const uint32_t cw = first ? 0u : my_carry[(size_t)k * 256];
            uint32_t cout = 0;
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t d = w[i];
#pragma unroll 1
                for (int bb = 0; bb < 4; ++bb) {
                    const int b = i * 4 + bb;
                    const unsigned co = literal_step<W>((const lds_u8*)smem + (d & 0xFFu), X, (cw >> b) & 1u);
                    cout |= co << b;
                    d >>= 8;
                }
            }
            if (!last)
                my_carry[(size_t)k * 256] = (uint16_t)cout;
        }
#pragma unroll
        for (int j = 0; j < 2 * W; ++j)
            res += __popc(~X[j]);
    }
    return res;
}

template <bool QUIRK, bool FUSE>
__global__ __launch_bounds__(256) void lcs_long_kernel(RowsArgs a, uint16_t* carry, int n_chunks_max)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[]; // SEGW x 256 bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int R = a.refs_per_block;
    int bx, by;
    block_coords(a, bx, by);
    const int ref0 = by * R;
    const int nr = min(R, a.n_refs - ref0);
    const int c0 = bx * 256;
    if (block_is_above_diagonal(a, ref0, nr, c0))
        return;

    const int c = c0 + tid;
    const bool valid = c < a.n_cols;
    [[maybe_unused]] lds_u64* f_col = (lds_u64*)(smem + (size_t)SEGW * 256);
    [[maybe_unused]] lds_u64* f_row = f_col + 256;
    if constexpr (FUSE) fuse_init(a, f_col, f_row, ref0, nr, c, valid, tid); // the first segment pass starts with a barrier
    const int pid = a.col_ids ? a.col_ids[valid ? c : 0] : a.col_begin + (valid ? c : 0);
    const uint32_t len_p = valid ? a.lens[pid] : 0u;
    const uint8_t* pbase = a.tiles + a.tile_base[pid >> 6] + (uint64_t)(pid & 63) * 16;
    const int my_chunks = (int)((len_p + 15) >> 4);
    const int wave_chunks = wave_max(my_chunks);
    uint16_t* my_carry = carry + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * n_chunks_max) * 256 + tid;

    for (int r = 0; r < nr; ++r) {
        const int rid = a.ref_ids ? a.ref_ids[ref0 + r] : a.ref_begin + ref0 + r;
        const int n_words = (int)((a.lens[rid] + 63) / 64);
        uint32_t res = 0;
        for (int word0 = 0; word0 < n_words;) {
            const int left = n_words - word0;
            const bool first = word0 == 0;
#define LCS_LONG_PASS(W)                                                                                       \
    res += long_segment_pass<QUIRK, W>(a, rid, word0, first, left <= W, smem, pbase, my_chunks, wave_chunks,  \
                                       my_carry, wave, lane);                                                  \
    word0 += W;
            if (SEGW >= 32 && left > 24) { LCS_LONG_PASS(32) }
            else if (SEGW >= 24 && left > 16) { LCS_LONG_PASS(24) }
            else if (left > 8) { LCS_LONG_PASS(16) }
            else { LCS_LONG_PASS(8) }
#undef LCS_LONG_PASS
        }
        if (valid && (!FUSE || a.out)) store_result(a, ref0 + r, c, res);
        if constexpr (FUSE) {
            const uint32_t one[1] = {res};
            fuse_group<1>(a, f_col, f_row, ref0, r, 1, c, a.n_cols, one, tid);
        }
    }
    if constexpr (FUSE) {
        __syncthreads();
        fuse_flush(a, f_col, f_row, ref0, nr, c, valid, tid);
    }
}

// ---- host-side dispatch ---------------------------------------------------------------

// how this translation unit's device code was built (csrc/Makefile): "on" = the register-renaming pass ran and its
// equivalence check passed, "off" = RECOLOR=0, "failed" = the pass or the check failed and the kernels are as compiled
#ifndef LCSGPU_RECOLOR_STATE
#define LCSGPU_RECOLOR_STATE "off"
#endif
// the first 10 hex digits of the sha256 of the listing this unit's device code was assembled from: two libraries with
// the same id run the same LCS kernels, instruction for instruction (profiles are matched to a library by it)
#ifndef LCSGPU_KERNEL_ID
#define LCSGPU_KERNEL_ID "unknown"
#endif
#ifndef LCS_FUSED_TU
const char* recolor_state() { return LCSGPU_RECOLOR_STATE; }
const char* kernel_id() { return LCSGPU_KERNEL_ID; }

int h_class(uint32_t len)
{
    // half-word counts that have their own instantiation; others round up (extra half-words
    // are all-ones no-ops: they cost time but cannot change the result)
    const int h = (int)((len + 31) / 32);
    if (h <= 1) return 1;
    if (h <= 32) return h;
    if (h <= 64) return (h + 1) & ~1;
    return 0; // long path
}

int quirk_h_class(uint32_t len)
{
    const int h = (int)((len + 31) / 32);
    if (h <= 8) return 8;
    if (h <= 16) return 16;
    if (h <= 32) return 32;
    if (h <= 64) return 64;
    return 0;
}

#endif // !LCS_FUSED_TU
#ifndef LCS_RG13
#define LCS_RG13 4 // measurement aid: refs advanced together in the 13-half-word (400 aa) instantiation
#endif
#ifndef LCS_FUSED_TU
static int rg_of(int h) { return h == 13 ? LCS_RG13 : h <= 16 ? 4 : (h <= 32 ? 2 : 1); }

int refs_per_block(int h, bool quirk, bool fused)
{
    if (h == 0) return 8; // long path: refs are processed one after another
    const int rg = quirk ? 1 : rg_of(h);
    const int w = (h + 1) / 2;
    // 32 KB of LDS per workgroup = 5 workgroups per CU; a fused launch keeps its records (3 KB) inside that budget
    int r = (int)((32 * 1024 - (fused ? FUSE_LDS_BYTES : 0)) / (size_t)(w * 256));
    if (r > 32) r = 32;
    r = r / rg * rg;
    if (r < rg) r = rg;
    return r;
}

// Small launches: a workgroup with the full complement of refs runs for hundreds of microseconds,
// and a few hundred such workgroups leave most of the chip idle for that long.  Fewer refs per
// workgroup (a multiple of the register group) give more, shorter workgroups; the full complement
// is kept once the launch has enough of them anyway.
int refs_per_block_for(int h, bool quirk, long n_refs, long col_blocks, bool fused)
{
    const int full = refs_per_block(h, quirk, fused);
    if (h == 0) return full;
    const int rg = quirk ? 1 : rg_of(h);
    const long want = 2048; // workgroups
    int r = full;
    while (r > rg && ((n_refs + r - 1) / r) * col_blocks < want) r = std::max(rg, r / 2 / rg * rg);
    return r;
}
#endif // !LCS_FUSED_TU

template <int H, int RG, bool QUIRK, bool FUSE>
static hipError_t launch_one(const RowsArgs& a, dim3 grid, hipStream_t stream)
{
    const size_t lds = (size_t)a.refs_per_block * ((H + 1) / 2) * 256 + (FUSE ? FUSE_LDS_BYTES : 0);
    if constexpr (QUIRK) {
        if (a.block_threads != 0 && a.block_threads != 256) return hipErrorInvalidValue;
        hipLaunchKernelGGL((lcs_rows_kernel_quirk<H, FUSE>), grid, dim3(256), lds, stream, a);
    } else {
        if (a.block_threads != 0 && a.block_threads != 256 && (a.block_threads != 64 || !a.jobs || FUSE)) return hipErrorInvalidValue;
        hipLaunchKernelGGL((lcs_rows_kernel_pipe<H, RG, LCS_LOOKAHEAD, FUSE>), grid, dim3(a.block_threads == 64 ? 64 : 256), lds, stream, a);
    }
    return hipGetLastError();
}

template <bool FUSE>
static hipError_t launch_rows_t(int h, bool quirk, const RowsArgs& a, int grid_x, int grid_y, hipStream_t stream)
{
    const dim3 grid((unsigned)grid_x, (unsigned)grid_y);
    if (quirk) {
        switch (h) {
        case 8: return launch_one<8, 1, true, FUSE>(a, grid, stream);
        case 16: return launch_one<16, 1, true, FUSE>(a, grid, stream);
        case 32: return launch_one<32, 1, true, FUSE>(a, grid, stream);
        case 64: return launch_one<64, 1, true, FUSE>(a, grid, stream);
        default: return hipErrorInvalidValue;
        }
    }
    switch (h) {
#define LCS_CASE(B, G) case B: return launch_one<B, G, false, FUSE>(a, grid, stream);
        LCS_CASE(1, 4) LCS_CASE(2, 4) LCS_CASE(3, 4) LCS_CASE(4, 4) LCS_CASE(5, 4) LCS_CASE(6, 4)
        LCS_CASE(7, 4) LCS_CASE(8, 4) LCS_CASE(9, 4) LCS_CASE(10, 4) LCS_CASE(11, 4) LCS_CASE(12, 4)
        LCS_CASE(13, LCS_RG13) LCS_CASE(14, 4) LCS_CASE(15, 4) LCS_CASE(16, 4)
        LCS_CASE(17, 2) LCS_CASE(18, 2) LCS_CASE(19, 2) LCS_CASE(20, 2) LCS_CASE(21, 2) LCS_CASE(22, 2)
        LCS_CASE(23, 2) LCS_CASE(24, 2) LCS_CASE(25, 2) LCS_CASE(26, 2) LCS_CASE(27, 2) LCS_CASE(28, 2)
        LCS_CASE(29, 2) LCS_CASE(30, 2) LCS_CASE(31, 2) LCS_CASE(32, 2)
        LCS_CASE(34, 1) LCS_CASE(36, 1) LCS_CASE(38, 1) LCS_CASE(40, 1) LCS_CASE(42, 1) LCS_CASE(44, 1)
        LCS_CASE(46, 1) LCS_CASE(48, 1) LCS_CASE(50, 1) LCS_CASE(52, 1) LCS_CASE(54, 1) LCS_CASE(56, 1)
        LCS_CASE(58, 1) LCS_CASE(60, 1) LCS_CASE(62, 1) LCS_CASE(64, 1)
#undef LCS_CASE
    default: return hipErrorInvalidValue;
    }
}

template <bool FUSE>
static hipError_t launch_long_t(bool quirk, const RowsArgs& a, int grid_x, int grid_y, void* carry, int n_chunks_max,
                                hipStream_t stream)
{
    const dim3 grid((unsigned)grid_x, (unsigned)grid_y);
    const size_t lds = (size_t)SEGW * 256 + (FUSE ? FUSE_LDS_BYTES : 0);
    if (quirk)
        hipLaunchKernelGGL((lcs_long_kernel<true, FUSE>), grid, dim3(256), lds, stream, a, (uint16_t*)carry, n_chunks_max);
    else
        hipLaunchKernelGGL((lcs_long_kernel<false, FUSE>), grid, dim3(256), lds, stream, a, (uint16_t*)carry, n_chunks_max);
    return hipGetLastError();
}

// This source is compiled twice (csrc/Makefile): as the plain translation unit -- every kernel without the fold, the
// host-side planning helpers -- and, with LCS_FUSED_TU, as the unit that holds the FUSE instantiations only.  Two
// units compile (and get their register pass) side by side.
#ifdef LCS_FUSED_TU
const char* recolor_state_fused() { return LCSGPU_RECOLOR_STATE; }
const char* kernel_id_fused() { return LCSGPU_KERNEL_ID; }
hipError_t launch_rows_fused(int h, bool quirk, const RowsArgs& a, int grid_x, int grid_y, hipStream_t stream)
{
    return launch_rows_t<true>(h, quirk, a, grid_x, grid_y, stream);
}
hipError_t launch_long_fused(bool quirk, const RowsArgs& a, int grid_x, int grid_y, void* carry, int n_chunks_max,
                             hipStream_t stream)
{
    return launch_long_t<true>(quirk, a, grid_x, grid_y, carry, n_chunks_max, stream);
}
#else
hipError_t launch_rows_fused(int h, bool quirk, const RowsArgs& a, int grid_x, int grid_y, hipStream_t stream);
hipError_t launch_long_fused(bool quirk, const RowsArgs& a, int grid_x, int grid_y, void* carry, int n_chunks_max,
                             hipStream_t stream);
hipError_t launch_rows(int h, bool quirk, const RowsArgs& a, int grid_x, int grid_y, hipStream_t stream)
{
    if (a.fuse.on) return launch_rows_fused(h, quirk, a, grid_x, grid_y, stream);
    return launch_rows_t<false>(h, quirk, a, grid_x, grid_y, stream);
}

size_t long_carry_bytes(int grid_x, int grid_y, int n_chunks_max)
{
    return (size_t)grid_x * grid_y * (size_t)n_chunks_max * 256 * sizeof(uint16_t);
}

hipError_t launch_long(bool quirk, const RowsArgs& a, int grid_x, int grid_y, void* carry, int n_chunks_max,
                       hipStream_t stream)
{
    if (a.fuse.on) return launch_long_fused(quirk, a, grid_x, grid_y, carry, n_chunks_max, stream);
    return launch_long_t<false>(quirk, a, grid_x, grid_y, carry, n_chunks_max, stream);
}
#endif

} // namespace lcsgpu
