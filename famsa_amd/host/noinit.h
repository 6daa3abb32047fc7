// noinit.h -- an allocator whose vectors do not fill what a resize adds: for buffers whose every element is written before
// it is read (the reader's bytes, LCS values the engine delivers, per-level scratch of the tree heuristics).  Zeros first
// would be a second pass over the memory -- and the first touch of its pages -- on one thread.
#pragma once
#include <memory>
#include <utility>

namespace famsa_host {

template <class T>
struct NoInit : std::allocator<T> {
    template <class U> struct rebind { using other = NoInit<U>; };
    NoInit() = default;
    template <class U> NoInit(const NoInit<U>&) {}
    template <class U> void construct(U* p) noexcept { ::new ((void*)p) U; } // default-init: nothing for a byte
    template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};

} // namespace famsa_host
