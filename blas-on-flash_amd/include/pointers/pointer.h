// pointer.h -- flash_ptr<T>: a typed {address tag, byte offset, file handle} triple
// naming a location inside a file (reference include/pointers/pointer.h:14-75).
// Public members, arithmetic in elements, converting between element types, and
// hashing/equality on the address tag only -- exactly the reference's contract.
#pragma once

#include <cstdint>
#include <functional>
#include <string>
#include <type_traits>

#include "bof_types.h"
#include "file_handles/file_handle.h"

namespace flash {
  template<typename T>
  struct flash_ptr {
    T* ptr;               // unique per byte of the mapped file (mmap address)
    FBLAS_UINT foffset;   // byte offset from the start of the file
    BaseFileHandle* fop;  // non-owning

    flash_ptr() : ptr(nullptr), foffset(0), fop(nullptr) {}
    flash_ptr(T* p, FBLAS_UINT byte_off, BaseFileHandle* handle)
        : ptr(p), foffset(byte_off), fop(handle) {}

    // advance by n_vals ELEMENTS (address tag and byte offset move together)
    flash_ptr operator+(FBLAS_UINT n_vals) const {
      return flash_ptr<T>(ptr + n_vals, foffset + n_vals * sizeof(T), fop);
    }

    template<typename X>
    bool operator==(const flash_ptr<X>& o) const {
      return static_cast<const void*>(ptr) == static_cast<const void*>(o.ptr) &&
             foffset == o.foffset && fop == o.fop;
    }

    T* get_raw_ptr() const { return ptr; }

    template<class Q = T>
    typename std::enable_if<!std::is_same<Q, void>::value, T>::type& operator*() {
      return *ptr;
    }

    // reinterpret as a pointer to another element type (same byte position)
    template<typename W>
    operator flash_ptr<W>() const {
      return flash_ptr<W>((W*) ptr, foffset, fop);
    }

    operator std::string() const {
      return "[" + std::to_string((uint64_t) fop) + "-" + std::to_string(foffset) + "]";
    }
  };

  struct FlashPtrHasher {
    size_t operator()(flash_ptr<void> const& key) const {
      return std::hash<void*>()(key.get_raw_ptr());
    }
  };
  struct FlashPtrEq {
    bool operator()(flash_ptr<void> const& a, flash_ptr<void> const& b) const {
      return a.get_raw_ptr() == b.get_raw_ptr();
    }
  };
}  // namespace flash
