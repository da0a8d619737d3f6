// rccl_transport.cpp — see include/multih_rccl.h.
#include "multih_rccl.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

#include <unistd.h>

struct mhr_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    long long calls = 0;
};

static thread_local std::string g_err;
static_assert(sizeof(ncclUniqueId) == MHR_ID_BYTES, "RCCL's unique id is 128 bytes");

static int fail(const std::string& what, const char* detail)
{
    g_err = what + ": " + (detail ? detail : "?");
    return 1;
}

extern "C" {

const char* mhr_last_error(void) { return g_err.c_str(); }

int mhr_unique_id(unsigned char id[MHR_ID_BYTES])
{
    ncclUniqueId u;
    const ncclResult_t r = ncclGetUniqueId(&u);
    if (r != ncclSuccess) return fail("ncclGetUniqueId", ncclGetErrorString(r));
    std::memcpy(id, &u, MHR_ID_BYTES);
    return 0;
}

int mhr_init(mhr_comm** out, int rank, int world, const unsigned char id[MHR_ID_BYTES], int device)
{
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return fail("mhr_init", "bad argument");
    *out = nullptr;
    hipError_t he = hipSetDevice(device);
    if (he != hipSuccess) return fail("hipSetDevice", hipGetErrorString(he));
    ncclUniqueId u;
    std::memcpy(&u, id, MHR_ID_BYTES);
    mhr_comm* c = new mhr_comm();
    c->rank = rank; c->world = world; c->device = device;
    const ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) { delete c; return fail("ncclCommInitRank", ncclGetErrorString(r)); }
    *out = c;
    return 0;
}

int mhr_init_from_file(mhr_comm** out, int rank, int world, const char* path, int device, int timeout_s)
{
    if (!path) return fail("mhr_init_from_file", "null path");
    unsigned char id[MHR_ID_BYTES];
    if (rank == 0) {
        if (mhr_unique_id(id)) return 1;
        // a private temporary next to `path`, created exclusively (never through a planted link), then renamed over it
        std::string tmp = std::string(path) + ".XXXXXX";
        const int fd = mkstemp(&tmp[0]);
        if (fd < 0) return fail("mkstemp", tmp.c_str());
        const bool ok = write(fd, id, MHR_ID_BYTES) == (ssize_t)MHR_ID_BYTES;
        close(fd);
        if (!ok) { unlink(tmp.c_str()); return fail("write", tmp.c_str()); }
        if (std::rename(tmp.c_str(), path) != 0) return fail("rename", path);      // readers see the whole id or nothing
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            FILE* f = std::fopen(path, "rb");
            if (f) {
                const size_t got = std::fread(id, 1, MHR_ID_BYTES, f);
                std::fclose(f);
                if (got == MHR_ID_BYTES) break;
            }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s)) return fail("waiting for the unique id", path);
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    }
    return mhr_init(out, rank, world, id, device);
}

int mhr_allgather(void* comm, const void* send_dev, void* recv_dev, unsigned long long bytes_per_rank, void* hip_stream)
{
    mhr_comm* c = static_cast<mhr_comm*>(comm);
    if (!c || !c->comm) return fail("mhr_allgather", "no communicator");
    const ncclResult_t r = ncclAllGather(send_dev, recv_dev, (size_t)bytes_per_rank, ncclChar, c->comm, (hipStream_t)hip_stream);
    if (r != ncclSuccess) return fail("ncclAllGather", ncclGetErrorString(r));
    ++c->calls;
    return 0;
}

long long mhr_calls(const mhr_comm* c) { return c ? c->calls : 0; }

int mhr_count(const mhr_comm* c)
{
    if (!c || !c->comm) { fail("mhr_count", "no communicator"); return -1; }
    int n = -1;
    const ncclResult_t r = ncclCommCount(c->comm, &n);
    if (r != ncclSuccess) { fail("ncclCommCount", ncclGetErrorString(r)); return -1; }
    return n;
}

int mhr_rank(const mhr_comm* c)
{
    if (!c || !c->comm) { fail("mhr_rank", "no communicator"); return -1; }
    int n = -1;
    const ncclResult_t r = ncclCommUserRank(c->comm, &n);
    if (r != ncclSuccess) { fail("ncclCommUserRank", ncclGetErrorString(r)); return -1; }
    return n;
}

int mhr_version(void)
{
    int v = -1;
    const ncclResult_t r = ncclGetVersion(&v);
    if (r != ncclSuccess) { fail("ncclGetVersion", ncclGetErrorString(r)); return -1; }
    return v;
}

void mhr_destroy(mhr_comm* c)
{
    if (!c) return;
    if (c->comm) (void)ncclCommDestroy(c->comm);
    delete c;
}

}
