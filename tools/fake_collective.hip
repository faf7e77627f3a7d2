// Stand-ins for RCCL's collectives (same C signatures as ncclAllReduce / ncclReduceScatter / ncclAllGather) for
// ONE-GPU boxes, where RCCL cannot run more than one rank.  Test / measurement infrastructure, not product code.
//
//   fake_*  TIMING MODEL (tools/ddp_model.py): exchanges nothing; occupies `blocks` workgroups on the given stream
//           for  latency + algorithmic link bytes / bus bandwidth  (a bounded wait on the constant-rate wall clock, so
//           it always exits) -- what a collective does to the step around it: it takes time on its stream and CUs
//           away from the kernels beside it.  Link bytes of an all-reduce of n bytes over w ranks: 2 (w-1)/w n;
//           of a reduce-scatter / all-gather whose per-rank piece is n bytes: (w-1) n.
//   shm_*   FUNCTIONAL (tests/test_ddp_gpu.py): a real exchange between PROCESSES that share one GPU, through a POSIX
//           shared-memory segment and the host: synchronise the stream, copy the operand out, meet the other ranks at
//           a barrier, reduce, copy the result back.  Slow and synchronous -- it is there so that rv_plan_step_ddp's
//           bucket arithmetic, 1/world scaling, shard ownership and stream ordering run with world > 1 and rank > 0
//           before the first multi-GPU job does.  Every wait is bounded (30 s) and returns an error code.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <new>
#include <vector>

// ------------------------------------------------------------------------------------------------ timing model
struct FakeComm {
  int blocks, threads, lds_bytes, world;
  float latency_us;     // start-up cost of one collective
  float bus_gb_per_s;   // per-GPU link bandwidth the algorithmic bytes move at
};

__global__ void k_hold(long long ticks) {
  extern __shared__ char lds[];
  if (threadIdx.x == 0) lds[0] = 1;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

static int hold(const FakeComm* c, double link_bytes, void* stream) {
  if (c->blocks <= 0 || c->bus_gb_per_s <= 0.f) return 0;
  const double us = c->latency_us + link_bytes / (c->bus_gb_per_s * 1e3);
  // wall_clock64 ticks at 100 MHz on this part
  hipLaunchKernelGGL(k_hold, dim3(c->blocks), dim3(c->threads), c->lds_bytes, (hipStream_t)stream, (long long)(us * 100.0));
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

static size_t dtype_bytes(int dtype) {   // ncclInt8 0, ncclUint8 1, ncclFloat16 6, ncclFloat32 7, ncclBfloat16 9
  return dtype == 7 ? 4 : (dtype == 6 || dtype == 9) ? 2 : dtype <= 1 ? 1 : 4;
}

extern "C" int fake_allreduce(const void*, void*, size_t count, int dtype, int, void* comm, void* stream) {
  const FakeComm* c = (const FakeComm*)comm;
  const double w = c->world > 1 ? c->world : 1;
  return hold(c, 2.0 * (w - 1.0) / w * (double)count * dtype_bytes(dtype), stream);
}

// ------------------------------------------------------------------------------------------------ functional
struct ShmHdr {
  std::atomic<int> arrived;
  std::atomic<int> generation;
  std::atomic<int> failed;
};
struct ShmComm {
  int world, rank;
  size_t cap;            // bytes per rank slot
  ShmHdr* hdr;
  char* slots;           // world * cap
  size_t map_bytes;
  char name[128];
  std::vector<char> tmp;
  int bf16_ring = 0;     // shm_set_bf16_ring
};

static double now_s() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}

// sense-reversing barrier over the ranks; 0 = met, 1 = timed out or another rank failed
static int shm_barrier(ShmComm* c) {
  ShmHdr* h = c->hdr;
  const int gen = h->generation.load(std::memory_order_acquire);
  if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->world) {
    h->arrived.store(0, std::memory_order_relaxed);
    h->generation.store(gen + 1, std::memory_order_release);
    return h->failed.load() ? 1 : 0;
  }
  const double t0 = now_s();
  while (h->generation.load(std::memory_order_acquire) == gen) {
    if (h->failed.load() || now_s() - t0 > 30.0) { h->failed.store(1); return 1; }
    usleep(50);
  }
  return h->failed.load() ? 1 : 0;
}

extern "C" void* shm_comm_create(const char* name, int world, int rank, size_t cap_bytes) {
  if (!name || world < 1 || rank < 0 || rank >= world || strlen(name) >= 120) return nullptr;
  const size_t hdr = 4096, total = hdr + (size_t)world * cap_bytes;
  int fd = -1;
  if (rank == 0) {
    shm_unlink(name);
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)total) != 0) return nullptr;
  } else {
    const double t0 = now_s();
    for (;;) {   // rank 0 creates and sizes the segment; the others wait for its full size
      fd = shm_open(name, O_RDWR, 0600);
      struct stat st;
      if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size == total) break;
      if (fd >= 0) close(fd);
      if (now_s() - t0 > 30.0) return nullptr;
      usleep(1000);
    }
  }
  void* m = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return nullptr;
  ShmComm* c = new (std::nothrow) ShmComm();
  if (!c) return nullptr;
  c->world = world; c->rank = rank; c->cap = cap_bytes; c->map_bytes = total;
  c->hdr = (ShmHdr*)m; c->slots = (char*)m + hdr;
  strcpy(c->name, name);
  c->tmp.resize(cap_bytes);
  // a fresh segment is zero-filled: arrived = generation = failed = 0
  if (shm_barrier(c)) { munmap(m, total); delete c; return nullptr; }
  return c;
}

// 1: bf16 all-reduces / reduce-scatters accumulate hop by hop in bf16, in ring order (see reduce_into); 0 (default): in fp32
extern "C" void shm_set_bf16_ring(void* comm, int on) {
  if (comm) ((ShmComm*)comm)->bf16_ring = on ? 1 : 0;
}

extern "C" void shm_comm_destroy(void* comm) {
  ShmComm* c = (ShmComm*)comm;
  if (!c) return;
  munmap(c->hdr, c->map_bytes);
  if (c->rank == 0) shm_unlink(c->name);
  delete c;
}

static float bf16_to_f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f_to_bf16(float f) {   // round to nearest even (no NaNs in these tests)
  uint32_t u; memcpy(&u, &f, 4);
  return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// out[i] = sum over ranks of slot_r[off + i], i in [0, n): fp32 in rank order.  bf16: accumulated in fp32 and rounded once
// (bf16_ring = 0: the kindest order a collective library could use), or -- bf16_ring = 1, shm_set_bf16_ring -- as a RING
// reduce-scatter does it: the element's chunk (n_total / world elements each) starts at rank (chunk + 1) % world and every
// hop adds the next rank's value and rounds the running sum to bf16 again: world - 1 roundings, the harshest order
// (RCCL's ring sums in the payload's type hop by hop).  `n_total`: the element count of the whole collective.
static void reduce_into(ShmComm* c, size_t elem_off, size_t n, int dtype, char* out, size_t n_total = 0) {
  if (dtype == 9 && c->bf16_ring && c->world > 1) {
    uint16_t* o = (uint16_t*)out;
    const size_t total = n_total ? n_total : n, chunk = (total + c->world - 1) / c->world;
    for (size_t i = 0; i < n; ++i) {
      const int start = (int)(((elem_off + i) / chunk + 1) % c->world);
      uint16_t run = ((const uint16_t*)(c->slots + start * c->cap))[elem_off + i];
      for (int k = 1; k < c->world; ++k) {
        const int r = (start + k) % c->world;
        run = f_to_bf16(bf16_to_f(run) + bf16_to_f(((const uint16_t*)(c->slots + r * c->cap))[elem_off + i]));
      }
      o[i] = run;
    }
    return;
  }
  if (dtype == 7) {
    float* o = (float*)out;
    for (size_t i = 0; i < n; ++i) {
      float s = 0.f;
      for (int r = 0; r < c->world; ++r) s += ((const float*)(c->slots + r * c->cap))[elem_off + i];
      o[i] = s;
    }
  } else {
    uint16_t* o = (uint16_t*)out;
    for (size_t i = 0; i < n; ++i) {
      float s = 0.f;
      for (int r = 0; r < c->world; ++r) s += bf16_to_f(((const uint16_t*)(c->slots + r * c->cap))[elem_off + i]);
      o[i] = f_to_bf16(s);
    }
  }
}

#define SHM_TRY(e) do { if ((e) != hipSuccess) { c->hdr->failed.store(1); return 2; } } while (0)

extern "C" int shm_allreduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, void* stream) {
  ShmComm* c = (ShmComm*)comm;
  const size_t bytes = count * dtype_bytes(dtype);
  if (op != 0 || (dtype != 7 && dtype != 9) || bytes > c->cap) return 3;
  SHM_TRY(hipStreamSynchronize((hipStream_t)stream));
  SHM_TRY(hipMemcpy(c->slots + c->rank * c->cap, send, bytes, hipMemcpyDeviceToHost));
  if (shm_barrier(c)) return 1;
  reduce_into(c, 0, count, dtype, c->tmp.data());
  if (shm_barrier(c)) return 1;   // everybody has read every slot before anybody writes one again
  SHM_TRY(hipMemcpy(recv, c->tmp.data(), bytes, hipMemcpyHostToDevice));
  return 0;
}


