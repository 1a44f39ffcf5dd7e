// fileio.h -- strided file <-> packed host buffer transfers (fileio.cpp)
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>
namespace bof {
// return 0 or -errno
int file_sread(int fd, uint64_t offset, uint64_t stride, uint64_t n_strides, uint64_t len,
               void *buf, bool use_aio);
int file_swrite(int fd, uint64_t offset, uint64_t stride, uint64_t n_strides, uint64_t len,
                const void *buf, bool use_aio);
// unaligned regions of an O_DIRECT descriptor, still through O_DIRECT (see fileio.cpp): the read puts
// the sector-aligned superset into buf (capacity >= len + 2 * file_dio_align, sector-aligned address) and
// sets *delta to where byte `off` landed; the write needs buf % page == off % page.
int file_read_widened(int fd, uint64_t off, uint64_t len, void *buf, uint64_t *delta, bool use_aio);
int file_write_split(int fd, uint64_t off, uint64_t len, const void *buf, bool use_aio);
bool file_is_direct(int fd);
uint64_t file_dio_align(int fd);  // O_DIRECT offset/length alignment of this file (>= 512)
// a buffered descriptor of the same file (fd itself unless it is O_DIRECT; cached; -1 on failure)
int file_buffered_fd(int fd);
// bytes that large buffered writes put into cached pages through a shared mapping (fileio.cpp)
uint64_t file_mapped_write_bytes();
void file_unmap_all();
void file_io_ops(uint64_t *reads, uint64_t *writes);  // requests issued so far (process-wide)
void file_forget(int fd);  // drop the cached buffered twin of fd (call before close)
void file_set_engine(int engine);   // 1 kernel AIO, 2 io_uring, anything else: $BOF_IO_ENGINE (read now) or AIO
// ---- io_uring engine (uring_io.cpp); BOF_IO_ENGINE=uring selects it for aligned O_DIRECT I/O ----
struct IoPiece { int fd; bool wr; void *buf; uint64_t len; uint64_t off; };
int uring_run(const std::vector<IoPiece> &pieces);   // 0 / -errno; -ENOSYS: no io_uring here
// pinned staging slots registered as io_uring FIXED buffers (PinnedRing does this)
void file_buffers_add(void *ptr, size_t bytes);
void file_buffers_remove(void *ptr);
void uring_op_counts(uint64_t *fixed, uint64_t *plain);
void uring_release_buffers();   // idle rings give up their fixed-buffer registrations
}  // namespace bof
