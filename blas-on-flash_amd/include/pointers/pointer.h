// pointer.h -- flash_ptr<T>: a typed {address tag, byte offset, file handle} triple
// naming a location inside a file.  Same public contract as the reference's
// include/pointers/pointer.h:14-75 (public members ptr/foffset/fop, element-wise
// `+`, comparison across element types, get_raw_ptr(), dereference, conversion to
// another element type, hashing/equality on the address tag only); written for this
// code base, where the tag is never dereferenced on the hot path: tiles travel
// file -> pinned ring -> HBM.
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
    T* ptr = nullptr;               // mmap address of the byte: unique per file position
    FBLAS_UINT foffset = 0;         // bytes from the start of the file
    BaseFileHandle* fop = nullptr;  // non-owning; map_file creates it, unmap_file deletes it

    flash_ptr() = default;
    // The reference declares this constructor only to reject it ("Bad usage", include/pointers/
    // pointer.h:22-24 -- an assert that release builds compile out); its kmeans driver still
    // names it for two pointers it never uses (drivers/kmeans.cpp:26-27).  Here: a null pointer.
    flash_ptr(T*) {}
    flash_ptr(T* tag, FBLAS_UINT byte_off, BaseFileHandle* handle)
        : ptr(tag), foffset(byte_off), fop(handle) {}

    T* get_raw_ptr() const { return ptr; }

    // same byte position seen as elements of another type
    template<typename W>
    operator flash_ptr<W>() const {
      return flash_ptr<W>(reinterpret_cast<W*>(const_cast<typename std::remove_const<T>::type*>(ptr)),
                          foffset, fop);
    }

    template<class Q = T>
    typename std::enable_if<!std::is_same<Q, void>::value, Q>::type& operator*() {
      return *ptr;
    }

    // "[<handle>-<offset>]", used in log lines
    operator std::string() const {
      std::string out("[");
      out += std::to_string(reinterpret_cast<uint64_t>(fop));
      out += "-";
      out += std::to_string(foffset);
      return out + "]";
    }
  };

  // advance by a number of ELEMENTS: tag and byte offset move together
  template<typename T>
  inline flash_ptr<T> operator+(const flash_ptr<T>& p, FBLAS_UINT n_vals) {
    return flash_ptr<T>(p.ptr + n_vals, p.foffset + n_vals * sizeof(T), p.fop);
  }

  // same file position, whatever the element types
  template<typename T, typename X>
  inline bool operator==(const flash_ptr<T>& a, const flash_ptr<X>& b) {
    const void* pa = a.ptr;
    const void* pb = b.ptr;
    return pa == pb && a.foffset == b.foffset && a.fop == b.fop;
  }

  // hash / equality functors keyed on the address tag alone (as the reference's maps are)
  struct FlashPtrHasher {
    size_t operator()(const flash_ptr<void>& key) const { return std::hash<const void*>()(key.ptr); }
  };
  struct FlashPtrEq {
    bool operator()(const flash_ptr<void>& a, const flash_ptr<void>& b) const { return a.ptr == b.ptr; }
  };
}  // namespace flash
