// adapter_check.cpp -- INTEGRATION.md section 2, COMPILED: the binding a FAMSA maintainer would add.
//
// TEST INFRASTRUCTURE (lives under oracle/, built into oracle/_ref/libfamsa_adapter.so, loaded only by
// tests/): it proves that the C-ABI of include/lcsgpu.h is a drop-in for the REFERENCE's own code, not only
// for this repository's re-implemented host layer.  GpuDistanceProvider offers the reference's
// batch-distance seam -- AbstractTreeGenerator::calculateDistanceMatrix / calculateDistanceVector
// (reference src/tree/AbstractTreeGenerator.hpp:130-182, 378-398; same template parameters and argument
// meaning, the CLCSBP& scratch argument replaced by the engine context) -- on top of liblcsgpu.so; the
// distances come out of the reference's own Transform functors (hpp:28-82).  GpuMSTPrim feeds the reference's own
// MSTPrim<D>::mst_to_dendogram (MSTPrim.cpp:784-833) with the edges of lcsgpu_mst_prim.  GpuUPGMA / GpuNJ derive from the
// reference's generators and override only the distance stage (UPGMA<D>::run, UPGMA.cpp:39-51;
// NeighborJoining<D>::run, NeighborJoining.cpp:10-23); the trees are built by the reference's own
// UPGMA<D>::computeTree (UPGMA.cpp:114-295) and NeighborJoining<D>::computeTree (NeighborJoining.cpp:33-118),
// i.e. by object code compiled from /root/reference (oracle/Makefile).  The orchestration around the
// generator (sort, duplicate removal, fromUnique, Newick) is ref_harness.cpp's, the one the reference's goldens pin.
#include "core/sequence.h"
#include "tree/AbstractTreeGenerator.hpp"
#include "tree/NeighborJoining.h"
#include "tree/UPGMA.h"

#include "tree/IPartialGenerator.h"
#include "lcs/lcsbp.h"

#include <algorithm>
#include <array>
#include <condition_variable>
#include <functional>
#include <iterator>
#include <limits>
#include <list>
#include <math.h>
#include <mutex>
#include <queue>
#include <stack>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>
// MSTPrim keeps its MST -> dendrogram step (mst_to_dendogram, tree/MSTPrim.cpp:784-833) and the edge record it takes
// private (members of a `class` without an access specifier); a maintainer would call it from inside the class.
// This test binding reaches it from a subclass instead: for this one header `class` reads `struct`, which changes
// the default access and nothing else (layout and mangled names are the same; every header it includes has been
// included above and is guarded).
#define class struct
#include "tree/MSTPrim.h"
#undef class

#include "../include/lcsgpu.h"

extern "C" long ref_tree_newick_with(void* h, int keep_dups, AbstractTreeGenerator* (*make)(void* user), void* user, char* out, long cap);

namespace {

class GpuDistanceProvider {
public:
    explicit GpuDistanceProvider(int device)
    {
        if (lcsgpu_create(device, &ctx_) != LCSGPU_OK) throw std::runtime_error(lcsgpu_last_error());
    }
    ~GpuDistanceProvider() { lcsgpu_destroy(ctx_); }

    // AbstractTreeGenerator::calculateDistanceMatrix (hpp:378-398): row i = ref sequences[i] against partners
    // sequences[0..i), lower triangle in TriangleMatrix::access order.
    template <class seq_type, class distance_type, typename Transform>
    void calculateDistanceMatrix(Transform& transform, seq_type* sequences, int n_seq, distance_type* out_matrix)
    {
        upload(sequences, n_seq);
        const size_t pairs = (size_t)n_seq * (n_seq - 1) / 2;
        std::vector<uint32_t> lcs(pairs);
        if (pairs && lcsgpu_lcs_triangle(ctx_, 0, n_seq, lcs.data(), 4) != LCSGPU_OK) throw std::runtime_error(lcsgpu_last_error());
        size_t k = 0;
        for (int i = 0; i < n_seq; ++i)
            for (int j = 0; j < i; ++j, ++k) // hpp:158: transform(lcs, ref->length, partner->length)
                out_matrix[k] = transform(lcs[k], ptr(sequences[i])->length, ptr(sequences[j])->length);
    }

    // AbstractTreeGenerator::calculateDistanceVector (hpp:130-182): one ref against a contiguous partner array
    // (the partners must be among the sequences of the last calculateDistanceMatrix / bind call).
    template <class seq_type, class distance_type, typename Transform>
    void calculateDistanceVector(Transform& transform, seq_type& ref, seq_type* sequences, int n_seqs, distance_type* out_vector)
    {
        std::vector<int32_t> cols(n_seqs);
        for (int j = 0; j < n_seqs; ++j) cols[j] = ptr(sequences[j])->sequence_no;
        const int32_t r = ptr(ref)->sequence_no;
        std::vector<uint32_t> lcs(n_seqs);
        if (n_seqs && lcsgpu_lcs_rect(ctx_, &r, 0, 1, cols.data(), 0, n_seqs, lcs.data(), n_seqs, 4) != LCSGPU_OK)
            throw std::runtime_error(lcsgpu_last_error());
        for (int j = 0; j < n_seqs; ++j) out_vector[j] = transform(lcs[j], ptr(ref)->length, ptr(sequences[j])->length);
    }

private:
    static CSequence* ptr(CSequence* s) { return s; }
    static CSequence* ptr(CSequence& s) { return &s; }

    // CSequence::data holds the symbol codes, padded to data_size with UNKNOWN_SYMBOL (core/sequence.h:28-32):
    // the engine wants the unpadded codes; ids = positions in `sequences` (= sequence_no after msa.cpp:559-561)
    template <class seq_type>
    void upload(seq_type* sequences, int n)
    {
        std::vector<uint64_t> offsets((size_t)n + 1, 0);
        for (int i = 0; i < n; ++i) offsets[i + 1] = offsets[i] + ptr(sequences[i])->length;
        std::vector<uint8_t> codes(offsets[n] ? offsets[n] : 1);
        for (int i = 0; i < n; ++i)
            for (uint32_t p = 0; p < ptr(sequences[i])->length; ++p) codes[offsets[i] + p] = (uint8_t)ptr(sequences[i])->data[p];
        if (lcsgpu_upload(ctx_, codes.data(), offsets.data(), n) != LCSGPU_OK) throw std::runtime_error(lcsgpu_last_error());
    }

    lcsgpu_ctx* ctx_ = nullptr;
};

template <Distance D>
class GpuUPGMA : public UPGMA<D> {
public:
    GpuUPGMA(bool modified, int device) : UPGMA<D>(1, instruction_set_t::none, modified), modified_(modified), gpu_(device) {}
    void run(std::vector<CSequence*>& sequences, tree_structure& tree) override
    {
        UPGMA_dist_t* distances = TriangleMatrix::allocate<UPGMA_dist_t>(sequences.size());
        Transform<UPGMA_dist_t, D> transform;
        gpu_.calculateDistanceMatrix<CSequence*, UPGMA_dist_t, decltype(transform)>(transform, sequences.data(), (int)sequences.size(), distances);
        if (modified_) this->template computeTree<true>(distances, (int)sequences.size(), tree);
        else this->template computeTree<false>(distances, (int)sequences.size(), tree);
        delete[] distances;
    }

private:
    bool modified_;
    GpuDistanceProvider gpu_;
};

template <Distance D>
class GpuNJ : public NeighborJoining<D> {
public:
    explicit GpuNJ(int device) : NeighborJoining<D>(1, instruction_set_t::none), gpu_(device) {}
    void run(std::vector<CSequence*>& sequences, tree_structure& tree) override
    {
        float* distances = TriangleMatrix::allocate<float>(sequences.size());
        Transform<float, D> transform;
        gpu_.calculateDistanceMatrix<CSequence*, float, decltype(transform)>(transform, sequences.data(), (int)sequences.size(), distances);
        this->computeTree(distances, (int)sequences.size(), tree);
        delete[] distances;
    }

private:
    GpuDistanceProvider gpu_;
};

// `-gt sl`, the reference's default (tree/TreeDefs.h:91), in its batched form: the whole of MSTPrim::run_view's
// distance work and key bookkeeping (tree/MSTPrim.cpp:356-533) is one lcsgpu_mst_prim call -- the LCS triangle and
// the MST are built in HBM -- and the reference's own mst_to_dendogram turns the n-1 edges, which arrive in the
// order Prim's algorithm adds them, into the tree.  mst_edges / v_prim_orders are filled exactly as run_view
// fills them (cpp:383-391: edge k gets prim order k, weight -d; the endpoint that is new gets the next order).
template <Distance D>
class GpuMSTPrim : public MSTPrim<D> {
public:
    explicit GpuMSTPrim(int device) : MSTPrim<D>(1, instruction_set_t::none), device_(device) {}
    void run(std::vector<CSequence*>& sequences, tree_structure& tree) override
    {
        const int n = (int)sequences.size();
        lcsgpu_ctx* ctx = nullptr;
        if (lcsgpu_create(device_, &ctx) != LCSGPU_OK) throw std::runtime_error(lcsgpu_last_error());
        struct Close {
            lcsgpu_ctx* c;
            ~Close() { lcsgpu_destroy(c); }
        } close{ctx};
        std::vector<uint64_t> offsets((size_t)n + 1, 0);
        for (int i = 0; i < n; ++i) offsets[i + 1] = offsets[i] + sequences[i]->length;
        std::vector<uint8_t> codes(offsets[n] ? offsets[n] : 1);
        for (int i = 0; i < n; ++i)
            for (uint32_t p = 0; p < sequences[i]->length; ++p) codes[offsets[i] + p] = (uint8_t)sequences[i]->data[p];
        if (lcsgpu_upload(ctx, codes.data(), offsets.data(), n) != LCSGPU_OK) throw std::runtime_error(lcsgpu_last_error());
        std::vector<lcsgpu_mst_edge> edges((size_t)std::max(n - 1, 1));
        const int kind = D == Distance::indel_div_lcs ? LCSGPU_DIST_INDEL_DIV_LCS : LCSGPU_DIST_INDEL075_DIV_LCS;
        if (lcsgpu_mst_prim(ctx, kind, edges.data()) != LCSGPU_OK) throw std::runtime_error(lcsgpu_last_error());

        std::vector<typename MSTPrim<D>::mst_edge_t> mst_edges;
        mst_edges.reserve(n);
        std::vector<int> v_prim_orders(n, n);
        int cur_prim_order = 0;
        v_prim_orders[0] = cur_prim_order++;
        for (int k = 0; k + 1 < n; ++k) {
            mst_edges.emplace_back(edges[k].from, edges[k].to, cur_prim_order, -edges[k].dist);
            if (v_prim_orders[edges[k].from] == n) v_prim_orders[edges[k].from] = cur_prim_order++;
            else v_prim_orders[edges[k].to] = cur_prim_order++;
        }
        this->mst_to_dendogram(mst_edges, v_prim_orders, tree);
    }

private:
    int device_;
};

struct Request {
    int gt, distance, device;
};

AbstractTreeGenerator* make_generator(void* user)
{
    const Request& r = *(const Request*)user;
    const bool d0 = r.distance == 0;
    switch (r.gt) { // ids as in ref_harness.cpp: 0 sl (MSTPrim), 2 upgma, 3 nj, 4 upgma_modified
    case 0:
        return d0 ? (AbstractTreeGenerator*)new GpuMSTPrim<Distance::indel_div_lcs>(r.device)
                  : (AbstractTreeGenerator*)new GpuMSTPrim<Distance::indel075_div_lcs>(r.device);
    case 2:
    case 4:
        return d0 ? (AbstractTreeGenerator*)new GpuUPGMA<Distance::indel_div_lcs>(r.gt == 4, r.device)
                  : (AbstractTreeGenerator*)new GpuUPGMA<Distance::indel075_div_lcs>(r.gt == 4, r.device);
    case 3:
        return d0 ? (AbstractTreeGenerator*)new GpuNJ<Distance::indel_div_lcs>(r.device)
                  : (AbstractTreeGenerator*)new GpuNJ<Distance::indel075_div_lcs>(r.device);
    default: throw std::runtime_error("this binding covers sl, upgma, upgma_modified and nj; slink, -dist_export and the "
                                      "heuristics run through the dispatcher seam (gpu_lcsbp.cpp)");
    }
}

} // namespace

// Newick text of `famsa -gt <sl|upgma|upgma_modified|nj> -gt_export`: the reference's generators over the GPU engine.
extern "C" long adapter_tree_newick(void* ref_handle, int gt, int distance, int keep_dups, int device, char* out, long cap)
{
    Request r{gt, distance, device};
    return ref_tree_newick_with(ref_handle, keep_dups, make_generator, &r, out, cap);
}
