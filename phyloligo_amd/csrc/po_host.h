// Host-only pieces of libphyloligo_amd.so (po_io.cpp): no HIP types, so the same translation unit also builds into the
// sanitizer harness (csrc/san/san_host_test.cpp, `make san`).
#pragma once

#include <stddef.h>
#include <stdint.h>

unsigned po_host_threads(unsigned cap);   // usable CPUs (affinity, cgroup quota), at most cap

// Rows that arrive chunk by chunk in one of two staging buffers -> the caller's (pageable) rows, copied out by a pool of
// host threads that lives for the whole call while the producer fills the other buffer.
//   issue(user, chunk, stage, first_row, n_rows)  start filling `stage` with rows [first_row, first_row + n_rows), row_bytes
//                                                 apart (asynchronous is fine)
//   wait(user)                                    return when the chunk issued last is complete in its staging buffer
// Both return 0 or a negative po_status; the first failure stops the ring and is returned.
struct po_ring_source {
    void* user;
    int (*issue)(void* user, uint64_t chunk, void* stage, uint64_t first_row, uint64_t n_rows);
    int (*wait)(void* user);
};
int po_ring_copy_rows(const po_ring_source& src, void* const stage[2], size_t stage_bytes, size_t row_bytes, uint64_t rows,
                      uint8_t* dst, size_t dst_pitch, unsigned n_threads);
