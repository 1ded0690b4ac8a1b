/* Test double of librccl for the CPU suite (tests/test_head_comm.py): the five NCCL entry points pv_comm binds,
 * exchanging HOST buffers between processes through POSIX shared memory.  It lets world-size-2 tests drive
 * pv_comm_unique_id / pv_comm_create / pv_comm_all_gather / pv_comm_destroy -- the exact C path `bench.py --gpus N`
 * runs -- without a GPU.  `stream` is ignored (calls complete synchronously).  Test infrastructure only. */
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define STUB_MAX_RANKS 8
#define STUB_SLOT (1u << 20) /* bytes per rank */

typedef struct { char internal[128]; } ncclUniqueId;

typedef struct {
  volatile int arrived;
  volatile int sense;
  volatile int inits;
  int calls[STUB_MAX_RANKS];
  unsigned char data[STUB_MAX_RANKS][STUB_SLOT];
} shared_t;

typedef struct {
  shared_t* sh;
  int rank, world, local_sense;
  char name[64];
} comm_t;

static void barrier(comm_t* c) {
  c->local_sense = !c->local_sense;
  if (__atomic_add_fetch(&c->sh->arrived, 1, __ATOMIC_ACQ_REL) == c->world) {
    __atomic_store_n(&c->sh->arrived, 0, __ATOMIC_RELEASE);
    __atomic_store_n(&c->sh->sense, c->local_sense, __ATOMIC_RELEASE);
  } else {
    while (__atomic_load_n(&c->sh->sense, __ATOMIC_ACQUIRE) != c->local_sense) usleep(50);
  }
}

int ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof(*id));
  struct timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  snprintf(id->internal, sizeof(id->internal), "/pv_rccl_stub_%d_%ld", (int)getpid(), (long)ts.tv_nsec);
  return 0;
}

int ncclCommInitRank(void** out, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > STUB_MAX_RANKS || rank < 0 || rank >= nranks) return 4; /* ncclInvalidArgument */
  comm_t* c = (comm_t*)calloc(1, sizeof(comm_t));
  strncpy(c->name, id.internal, sizeof(c->name) - 1);
  int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, sizeof(shared_t)) != 0) { free(c); return 2; /* ncclSystemError */ }
  c->sh = (shared_t*)mmap(NULL, sizeof(shared_t), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->sh == MAP_FAILED) { free(c); return 2; }
  c->rank = rank;
  c->world = nranks;
  if (rank == 0) printf("RCCL version : stub (the real library prints such a banner to the C stdout at its first communicator)\n");
  __atomic_add_fetch(&c->sh->inits, 1, __ATOMIC_ACQ_REL);
  while (__atomic_load_n(&c->sh->inits, __ATOMIC_ACQUIRE) < nranks) usleep(50);   /* rendezvous like the real one */
  *out = c;
  return 0;
}

int ncclAllGather(const void* send, void* recv, size_t count, int datatype, void* comm, void* stream) {
  (void)stream;
  comm_t* c = (comm_t*)comm;
  if (!c || datatype != 1 /* ncclUint8: pv_comm is byte-typed */ || count > STUB_SLOT) return 4;
  memcpy((void*)c->sh->data[c->rank], send, count);
  c->sh->calls[c->rank]++;
  barrier(c);
  for (int r = 0; r < c->world; ++r) memcpy((char*)recv + (size_t)r * count, (const void*)c->sh->data[r], count);
  barrier(c);
  return 0;
}

int ncclCommDestroy(void* comm) {
  comm_t* c = (comm_t*)comm;
  if (!c) return 4;
  if (c->rank == 0) shm_unlink(c->name);
  munmap(c->sh, sizeof(shared_t));
  free(c);
  return 0;
}

const char* ncclGetErrorString(int code) {
  switch (code) {
    case 0: return "no error";
    case 2: return "unhandled system error (stub)";
    case 4: return "invalid argument (stub)";
    default: return "stub error";
  }
}

/* ncclGetVersion: 2.21.5 unless PV_RCCL_STUB_VERSION says otherwise (the refusal test reports a major version 3) */
int ncclGetVersion(int* version) {
  const char* v = getenv("PV_RCCL_STUB_VERSION");
  *version = v ? atoi(v) : 22105;
  return 0;
}
