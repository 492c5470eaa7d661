// fake_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for the five RCCL entry points csrc/shardcomm.hip binds (ncclGetUniqueId,
// ncclCommInitRank, ncclAllGather, ncclCommDestroy, ncclGetErrorString), built as libfake_rccl.so and named to the library
// through AK_RCCL_PATH. RCCL refuses two ranks on one device and the builder / driver boxes have ONE GPU, so without it
// ak_index_search_sharded_dev -- the entry point INTEGRATION.md tells a ctypes-only maintainer to bind -- could only ever run at
// world size 1 (round-5 review, missing #4). Here several processes share the one GPU and exchange through a POSIX shared-memory
// segment: an all-gather is  stream-ordered D2H of the send buffer into this rank's slot -> barrier -> H2D of every slot into
// the receive buffer -> barrier, with the barriers timing out (ncclSystemError) instead of hanging when a rank never arrives --
// which is exactly what the failure-contract tests need to see.
//
// Nothing in archi_amd/ links or loads this file by itself; it implements no RCCL algorithm and makes no claim about RCCL's
// behaviour over xGMI. FAKE_RCCL_HOST_ONLY builds it without HIP (plain memcpy on host pointers) for the CPU suite's test of the
// segment / barrier logic (tests/test_sharded_cpu.py).
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>

#ifndef FAKE_RCCL_HOST_ONLY
#include <hip/hip_runtime_api.h>
#else
typedef void *hipStream_t;
#endif

extern "C" {

typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 };

}  // extern "C"

namespace {

constexpr uint64_t MAGIC = 0x616b66616b657231ull;      // "akfaker1"
constexpr size_t SLOT_BYTES = 8u << 20;                // per rank and collective: Q = 1024, k = 128 is 2.1 MB
constexpr int MAX_WORLD = 16;

struct Header {
    std::atomic<uint64_t> magic;
    std::atomic<int> world;
    std::atomic<int> arrived;        // sense-reversing barrier
    std::atomic<int> generation;
    std::atomic<int> attached;
    std::atomic<int> aborted;        // a rank timed out: everybody still waiting leaves with an error
    char pad[64];
};

struct Comm {
    Header *hdr = nullptr;
    char *slots = nullptr;
    size_t map_bytes = 0;
    int rank = 0, world = 1;
    char name[64] = {0};
    void *stage = nullptr;           // pinned staging buffer (device build)
    double timeout_s = 60.0;
};

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

// all `world` ranks arrive, or the call returns false after timeout_s (and poisons the segment so that nobody waits again)
bool barrier(Comm *c) {
    Header *h = c->hdr;
    if (h->aborted.load()) return false;
    const int gen = h->generation.load();
    if (h->arrived.fetch_add(1) + 1 == c->world) {
        h->arrived.store(0);
        h->generation.fetch_add(1);
        return true;
    }
    const double t0 = now_s();
    while (h->generation.load() == gen) {
        if (h->aborted.load()) return false;
        if (now_s() - t0 > c->timeout_s) {
            h->aborted.store(1);
            return false;
        }
        usleep(50);
    }
    return true;
}

}  // namespace

extern "C" {

const char *ncclGetErrorString(int e) {
    switch (e) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "fake_rccl: HIP call failed";
        case ncclSystemError: return "fake_rccl: a rank did not arrive within the timeout (or the segment could not be mapped)";
        case ncclInvalidArgument: return "fake_rccl: invalid argument (message larger than a slot, world > 16 ...)";
        default: return "fake_rccl: internal error";
    }
}

int ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    memset(id->internal, 0, sizeof(id->internal));
    unsigned r = 0;
    FILE *f = fopen("/dev/urandom", "rb");
    if (f) { if (fread(&r, sizeof(r), 1, f) != 1) r = (unsigned)time(nullptr); fclose(f); }
    snprintf(id->internal, sizeof(id->internal), "/akfake_%d_%08x", (int)getpid(), r);
    return ncclSuccess;
}

int ncclCommDestroy(ncclComm_t h);

int ncclCommInitRank(ncclComm_t *out, int world, ncclUniqueId id, int rank) {
    if (!out || world < 1 || world > MAX_WORLD || rank < 0 || rank >= world) return ncclInvalidArgument;
    id.internal[sizeof(id.internal) - 1] = 0;
    if (id.internal[0] != '/' || strlen(id.internal) >= sizeof(Comm::name)) return ncclInvalidArgument;
    Comm *c = new Comm();
    c->rank = rank; c->world = world;
    strcpy(c->name, id.internal);
    if (const char *t = getenv("FAKE_RCCL_TIMEOUT_S")) c->timeout_s = atof(t) > 0 ? atof(t) : c->timeout_s;
    c->map_bytes = sizeof(Header) + (size_t)world * SLOT_BYTES;
    int fd = -1;
    const double t0 = now_s();
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { if (fd >= 0) close(fd); delete c; return ncclSystemError; }
    } else {
        for (;;) {      // rank 0 creates the segment; the others wait for it to exist at its full size
            fd = shm_open(c->name, O_RDWR, 0600);
            struct stat sb;
            if (fd >= 0 && fstat(fd, &sb) == 0 && (size_t)sb.st_size >= c->map_bytes) break;
            if (fd >= 0) close(fd);
            fd = -1;
            if (now_s() - t0 > c->timeout_s) { delete c; return ncclSystemError; }
            usleep(200);
        }
    }
    void *p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->hdr = (Header *)p;
    c->slots = (char *)p + sizeof(Header);
    if (rank == 0) {
        c->hdr->world.store(world); c->hdr->arrived.store(0); c->hdr->generation.store(0); c->hdr->attached.store(0);
        c->hdr->aborted.store(0);
        c->hdr->magic.store(MAGIC);              // last: the others spin on it
    } else {
        while (c->hdr->magic.load() != MAGIC) {
            if (now_s() - t0 > c->timeout_s) { munmap(p, c->map_bytes); delete c; return ncclSystemError; }
            usleep(200);
        }
        if (c->hdr->world.load() != world) { munmap(p, c->map_bytes); delete c; return ncclInvalidArgument; }
    }
    c->hdr->attached.fetch_add(1);
#ifndef FAKE_RCCL_HOST_ONLY
    if (hipHostMalloc(&c->stage, (size_t)world * SLOT_BYTES) != hipSuccess) { munmap(p, c->map_bytes); delete c; return ncclUnhandledCudaError; }
#endif
    if (!barrier(c)) {                            // like ncclCommInitRank: returns when all ranks are in
        if (rank == 0) shm_unlink(c->name);
        ncclCommDestroy((ncclComm_t)c);
        return ncclSystemError;
    }
    if (rank == 0) shm_unlink(c->name);          // every rank has it mapped: the name can go (nothing is left behind in /dev/shm)
    *out = (ncclComm_t)c;
    return ncclSuccess;
}

int ncclCommDestroy(ncclComm_t h) {
    if (!h) return ncclSuccess;
    Comm *c = (Comm *)h;
#ifndef FAKE_RCCL_HOST_ONLY
    if (c->stage) (void)hipHostFree(c->stage);
#endif
    if (c->hdr) munmap((void *)c->hdr, c->map_bytes);
    delete c;
    return ncclSuccess;
}

// dtype: only the element SIZE matters to a gather (ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3, ncclInt64 4, ncclUint64 5,
// ncclFloat16 6, ncclFloat32 7, ncclFloat64 8, ncclBfloat16 9)
int ncclAllGather(const void *send, void *recv, size_t count, int dtype, ncclComm_t h, hipStream_t stream) {
    static const int esize[] = {1, 1, 4, 4, 8, 8, 2, 4, 8, 2};
    if (!h || !send || !recv || dtype < 0 || dtype > 9) return ncclInvalidArgument;
    Comm *c = (Comm *)h;
    const size_t bytes = count * (size_t)esize[dtype];
    if (bytes > SLOT_BYTES) return ncclInvalidArgument;
    char *mine = c->slots + (size_t)c->rank * SLOT_BYTES;
#ifndef FAKE_RCCL_HOST_ONLY
    // stream order: everything enqueued on `stream` before the collective has run when the send buffer is read
    if (hipMemcpyAsync(c->stage, send, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    memcpy(mine, c->stage, bytes);
#else
    (void)stream;
    memcpy(mine, send, bytes);
#endif
    if (!barrier(c)) return ncclSystemError;                 // every slot is written
#ifndef FAKE_RCCL_HOST_ONLY
    for (int r = 0; r < c->world; r++) memcpy((char *)c->stage + (size_t)r * bytes, c->slots + (size_t)r * SLOT_BYTES, bytes);
    if (hipMemcpyAsync(recv, c->stage, (size_t)c->world * bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
#else
    for (int r = 0; r < c->world; r++) memcpy((char *)recv + (size_t)r * bytes, c->slots + (size_t)r * SLOT_BYTES, bytes);
#endif
    if (!barrier(c)) return ncclSystemError;                 // every slot is read: the next collective may overwrite them
    return ncclSuccess;
}

}  // extern "C"
