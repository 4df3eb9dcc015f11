// bb_comm.cpp -- run-time binding of RCCL (see bb_comm.h).
#include "bb_comm.h"

#include <dlfcn.h>
#include <stdlib.h>

#include <mutex>

namespace bb {

const Rccl &rccl() {
    static Rccl table;
    static std::once_flag once;
    std::call_once(once, [] {
        // BB_NO_RCCL=1: behave as if librccl could not be loaded (a rehearsal of the paths
        // that run without the library's communicator: torch.distributed as the reference
        // of the exchange trial, and as the transport)
        if (const char *e = getenv("BB_NO_RCCL")) if (atoi(e) != 0) return;
        // RTLD_NOLOAD first: reuse the image the process already has (torch's)
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        table.GetUniqueId = (decltype(table.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        table.CommInitRank = (decltype(table.CommInitRank))dlsym(h, "ncclCommInitRank");
        table.CommDestroy = (decltype(table.CommDestroy))dlsym(h, "ncclCommDestroy");
        table.CommCount = (decltype(table.CommCount))dlsym(h, "ncclCommCount");
        table.CommAbort = (decltype(table.CommAbort))dlsym(h, "ncclCommAbort");
        table.AllReduce = (decltype(table.AllReduce))dlsym(h, "ncclAllReduce");
        table.GetErrorString = (decltype(table.GetErrorString))dlsym(h, "ncclGetErrorString");
        table.ok = table.GetUniqueId && table.CommInitRank && table.CommDestroy &&
                   table.AllReduce && table.GetErrorString;
    });
    return table;
}

}  // namespace bb
