/* TEST INFRASTRUCTURE (tests/test_comm_abi_two_ranks.py): a host-memory stand-in for the seven nccl* entry points libsuhpe.so resolves with dlopen
 * (csrc/comm.hip; selected with SUHPE_RCCL_LIB), so that the suhpe_comm_* C ABI can be driven by TWO ranks -- two processes -- on a box without
 * GPUs: argument order, counts, the average flag, roots, zero-length calls, error mapping.  It is NOT RCCL and never ships: buffers are host
 * pointers, the "stream" is ignored, every call is synchronous.  Ranks meet in a POSIX shared-memory segment named after the unique id.
 *
 *   gcc -O2 -shared -fPIC tests/host_build/hostmem_transport.c -o tests/host_build/libhostmem_transport.so -lrt -pthread
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define MAX_RANKS 8
#define SLOT_BYTES (8u << 20)           /* per-rank staging area: 2 Mi floats per call */

typedef struct { char bytes[128]; } UniqueId;

typedef struct {
    volatile int arrived;                /* barrier: counts arrivals of the current generation */
    volatile int generation;
    volatile int world;
    volatile int attached;
} Header;

typedef struct {
    Header* hdr;
    char* slots;                         /* MAX_RANKS x SLOT_BYTES */
    int rank, world;
    char name[64];
} CommImpl;

static void barrier(CommImpl* c) {
    Header* h = c->hdr;
    const int gen = h->generation;
    if (__sync_add_and_fetch(&h->arrived, 1) == c->world) {
        h->arrived = 0;
        __sync_synchronize();
        h->generation = gen + 1;
    } else {
        while (h->generation == gen) sched_yield();
    }
    __sync_synchronize();
}

int ncclGetUniqueId(UniqueId* id) {
    memset(id, 0, sizeof(*id));
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->bytes, sizeof(id->bytes), "/suhpe_hostmem_%d_%ld", (int)getpid(), (long)ts.tv_nsec);
    return 0;
}

int ncclCommInitRank(void** comm, int world, UniqueId id, int rank) {
    if (world < 1 || world > MAX_RANKS || rank < 0 || rank >= world) return 4;      /* ncclInvalidArgument */
    CommImpl* c = (CommImpl*)calloc(1, sizeof(CommImpl));
    c->rank = rank; c->world = world;
    strncpy(c->name, id.bytes, sizeof(c->name) - 1);
    const size_t total = 4096 + (size_t)MAX_RANKS * SLOT_BYTES;
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { free(c); return 2; }                                               /* ncclSystemError */
    if (ftruncate(fd, (off_t)total) != 0) { close(fd); free(c); return 2; }
    char* base = (char*)mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (base == MAP_FAILED) { free(c); return 2; }
    c->hdr = (Header*)base;
    c->slots = base + 4096;
    c->hdr->world = world;
    __sync_add_and_fetch(&c->hdr->attached, 1);
    while (c->hdr->attached < world) sched_yield();                                   /* every rank has mapped the segment */
    barrier(c);
    *comm = c;
    return 0;
}

int ncclCommDestroy(void* comm) {
    CommImpl* c = (CommImpl*)comm;
    if (!c) return 0;
    barrier(c);
    if (c->rank == 0) shm_unlink(c->name);
    munmap((void*)c->hdr, 4096 + (size_t)MAX_RANKS * SLOT_BYTES);
    free(c);
    return 0;
}

/* datatype 7 = float32 (the only one libsuhpe uses); op 0 = sum, 4 = avg */
int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, void* stream) {
    (void)stream;
    CommImpl* c = (CommImpl*)comm;
    if (dtype != 7 || (op != 0 && op != 4) || count * 4 > SLOT_BYTES) return 4;
    memcpy(c->slots + (size_t)c->rank * SLOT_BYTES, send, count * 4);
    barrier(c);
    float* out = (float*)recv;
    for (size_t i = 0; i < count; ++i) {
        float s = 0.f;
        for (int r = 0; r < c->world; ++r) s += ((const float*)(c->slots + (size_t)r * SLOT_BYTES))[i];      /* rank order: the same bits on every rank */
        out[i] = op == 4 ? s / (float)c->world : s;
    }
    barrier(c);
    return 0;
}

int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, void* stream) {
    (void)stream;
    CommImpl* c = (CommImpl*)comm;
    if (dtype != 7 || count * 4 > SLOT_BYTES) return 4;
    memcpy(c->slots + (size_t)c->rank * SLOT_BYTES, send, count * 4);
    barrier(c);
    for (int r = 0; r < c->world; ++r) memcpy((char*)recv + (size_t)r * count * 4, c->slots + (size_t)r * SLOT_BYTES, count * 4);
    barrier(c);
    return 0;
}

int ncclBroadcast(const void* send, void* recv, size_t count, int dtype, int root, void* comm, void* stream) {
    (void)stream;
    CommImpl* c = (CommImpl*)comm;
    if (dtype != 7 || root < 0 || root >= c->world || count * 4 > SLOT_BYTES) return 4;
    if (c->rank == root) memcpy(c->slots + (size_t)root * SLOT_BYTES, send, count * 4);
    barrier(c);
    memcpy(recv, c->slots + (size_t)root * SLOT_BYTES, count * 4);
    barrier(c);
    return 0;
}

const char* ncclGetErrorString(int rc) {
    switch (rc) {
        case 0: return "no error";
        case 2: return "system error (host-memory transport)";
        case 4: return "invalid argument (host-memory transport)";
        default: return "error (host-memory transport)";
    }
}
