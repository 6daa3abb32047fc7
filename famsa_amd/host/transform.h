// transform.h -- LCS length -> pairwise distance, bit-identical to the reference's Transform
// functors (reference tree/AbstractTreeGenerator.hpp:28-82).  The arithmetic type T and the
// operand order are part of the contract: MSTPrim / SLINK / -dist_export use T = double,
// UPGMA / NJ / -pid use T = float.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace famsa_host {

enum class Distance { indel_div_lcs = 0, indel075_div_lcs = 1, pairwise_identity = 2 };

// value returned when lcs == 0.  The reference writes nextafter(numeric_limits<T>::max(), 0)
// with an int second argument, so both operands are promoted to double: for T = double this is
// the double just below DBL_MAX, for T = float the (double) result rounds back to FLT_MAX.
template <class T>
inline T zero_lcs_distance()
{
    return (T)std::nextafter((double)std::numeric_limits<T>::max(), 0.0);
}

template <class T, Distance D>
struct Transform;

template <class T>
struct Transform<T, Distance::indel075_div_lcs> {
    std::vector<T> pow075; // (T) pow((double) indel, 0.75), grown on demand (hpp:43-48)
    T operator()(uint32_t lcs, uint32_t len1, uint32_t len2)
    {
        const uint32_t indel = len1 + len2 - 2 * lcs;
        if (indel >= pow075.size()) {
            size_t from = pow075.size();
            pow075.resize((size_t)indel + 1);
            for (size_t i = from; i <= indel; ++i) pow075[i] = (T)std::pow((double)(uint32_t)i, 0.75);
        }
        const T l = (T)lcs;
        if (l) return pow075[indel] / l;
        return zero_lcs_distance<T>();
    }
};

template <class T>
struct Transform<T, Distance::indel_div_lcs> {
    T operator()(uint32_t lcs, uint32_t len1, uint32_t len2)
    {
        const T indel = (T)(len1 + len2 - 2 * lcs);
        if (lcs) return indel / (T)lcs;
        return zero_lcs_distance<T>();
    }
};

template <class T>
struct Transform<T, Distance::pairwise_identity> {
    T operator()(uint32_t lcs, uint32_t len1, uint32_t len2) { return (T)lcs / (T)std::min(len1, len2); }
};

} // namespace famsa_host
