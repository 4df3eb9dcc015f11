// bb_comm.h -- RCCL, loaded at run time (dlopen), for the solver's per-iteration
// all-reduce.  No link-time dependency: single-GPU users never touch RCCL, and in
// a process that already has a librccl (torch bundles one) the same image is used.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>

namespace bb {

constexpr int kUniqueIdBytes = 128;  // NCCL_UNIQUE_ID_BYTES (rccl.h:40)

struct Rccl {
    // mirrors of the RCCL ABI we call (rccl.h:43, :187, :220, :260, :448, :466, :611)
    struct UniqueId { char internal[kUniqueIdBytes]; };
    typedef void *Comm;
    enum { kSum = 0, kFloat32 = 7, kFloat64 = 8, kSuccess = 0 };
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*CommCount)(const Comm, int *) = nullptr;
    int (*CommAbort)(Comm) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};

// The process-wide table (loaded once); ok == false if librccl is not loadable.
const Rccl &rccl();

}  // namespace bb
