// Per-instance frustum cull + compaction into VkDrawIndexedIndirectCommand records for gfx950.
// /root/reference/shaders/prep_indirect.comp:1-5 is an empty main(); the record layout and the
// "instances of a mesh are contiguous" rule come from /root/reference/src/vkmodel.cpp:79,96,108-110
// and /root/reference/inc/vkmodel.hpp:72,76-78.
//
// Compiled with -ffp-contract=off: the plane test is ((n.x*p.x + n.y*p.y) + n.z*p.z) + n.w >= 0 in
// binary32 with separate multiplies and adds, so the visible set is bit-identical to a CPU
// evaluation of the same expression.
//
// Both modes stream 32 bytes per instance (two perfectly coalesced float4 planes, 1 KiB per wave
// instruction) and are HBM-bound.  A wave owns 256 consecutive instances (4 per lane, all eight
// loads issued before the first use).
//   atomic mode        : ballot -> popcount -> one atomicAdd per (wave, mesh run) on the
//                        instanceCount field of the output command itself; lanes place their index
//                        by mbcnt rank.  Order inside a mesh segment depends on wave arrival.
//   deterministic mode : launch 1 writes one 64-bit visibility word per wave-row plus per-block
//                        counts; launch 2 places the indices in ascending order and writes the
//                        commands, each workgroup summing the counts of the workgroups before it
//                        (up to 8192 of them; beyond that a single-workgroup scan launch goes in
//                        between and the placement reads the scanned prefixes).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "vkr_kernels.h"

#ifndef VKR_ATOMIC_PER_LANE
#define VKR_ATOMIC_PER_LANE 4
#endif

namespace {

constexpr uint32_t kBlockThreads = 256;
constexpr uint32_t kPerLane = 4;
constexpr uint32_t kBlockInstances = kBlockThreads * kPerLane;  // 1024
constexpr uint32_t kWordsPerBlock = kBlockInstances / 64;       // 16

__device__ __forceinline__ bool aabb_visible(const float4 mn, const float4 mx, const float* __restrict__ pl) {
  bool vis = true;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float nx = pl[4 * k + 0], ny = pl[4 * k + 1], nz = pl[4 * k + 2], nw = pl[4 * k + 3];
    const float px = nx >= 0.0f ? mx.x : mn.x;
    const float py = ny >= 0.0f ? mx.y : mn.y;
    const float pz = nz >= 0.0f ? mx.z : mn.z;
    const float t0 = nx * px;
    const float t1 = ny * py;
    const float t2 = nz * pz;
    const float d = ((t0 + t1) + t2) + nw;
    vis = vis && (d >= 0.0f);
  }
  return vis;
}

// Streaming 16-byte load: the AABB planes are read exactly once.
__device__ __forceinline__ float4 load_stream(const float4* p) {
  // non-temporal: measured ~10 % faster than plain loads on the 2^22-instance stream
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t rank_in(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__global__ __launch_bounds__(256) void cull_init_cmds(const CullParams p) {
  const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= p.mesh_count) return;
  DrawCmd c = p.cmd_template[m];
  c.instanceCount = 0u;
  c.firstInstance = p.first_instance + p.mesh_base[m];
  p.cmd_out[m] = c;
}

constexpr uint32_t kAtomicPerLane = VKR_ATOMIC_PER_LANE;
constexpr uint32_t kAtomicChunk = 64u * kAtomicPerLane;  // instances per wave and loop trip
__device__ __forceinline__ bool any_pending(const uint64_t (&pend)[kAtomicPerLane]) {
  uint64_t all = 0;
#pragma unroll
  for (uint32_t k = 0; k < kAtomicPerLane; ++k) all |= pend[k];
  return all != 0ull;
}

// Loads and tests the instances of this lane.  Instance of (wave chunk, row k, lane l) is
// chunk*256 + k*64 + l, so every wave-level load covers 1 KiB of contiguous memory.
struct LaneWork {
  bool vis[kAtomicPerLane];
  uint32_t mesh[kAtomicPerLane];  // local mesh index (global id - first_mesh)
};

__device__ __forceinline__ LaneWork test_four(const CullParams& p, uint32_t chunk_base, uint32_t lane, bool& bad) {
  float4 mn[kAtomicPerLane], mx[kAtomicPerLane];
  bool in[kAtomicPerLane];
#pragma unroll
  for (uint32_t k = 0; k < kAtomicPerLane; ++k) {
    const uint32_t i = chunk_base + k * 64u + lane;
    in[k] = i < p.instance_count;
    const uint32_t ic = in[k] ? i : 0u;
    mn[k] = load_stream(p.aabb_min + ic);
    mx[k] = load_stream(p.aabb_max + ic);
  }
  LaneWork w;
#pragma unroll
  for (uint32_t k = 0; k < kAtomicPerLane; ++k) {
    w.mesh[k] = __float_as_uint(mn[k].w) - p.first_mesh;
    bool v = in[k] && aabb_visible(mn[k], mx[k], p.planes);
    if (v && w.mesh[k] >= p.mesh_count) {
      bad = true;
      v = false;
    }
    w.vis[k] = v;
  }
  return w;
}

// Counts accumulate in cmd_out[m].instanceCount, zeroed by cull_init_cmds in the launch before.
// Folding both into one launch was tried twice and is no faster than the 4.5 us the extra launch costs:
// (a) the first blocks initialise the records and bump a counter that every wave polls before its
// first atomic - the waves reach the poll before the initialising blocks have pushed their stores out
// (__threadfence) and stall for about as long as the launch took; with an agent-scope *acquire* in
// the poll it is 3x slower, because that invalidates the XCD's L2 once per wave (the L2s of the
// eight XCDs are not coherent with each other);  (b) a "last block" ticket at the end was 4x slower.
// Requesting the next chunk's boxes before working on the current one (so that the returning atomic waits
// alongside those loads) was measured too: 31-32 us per call against 30, at any grid size.  So was halving
// the number of atomics by letting a wave test two load groups of 256 consecutive instances one after the
// other before its atomic: 32 us (four groups: 40 us) - short-lived waves matter more than fewer atomics.
__global__ __launch_bounds__(256) void cull_atomic_kernel(const CullParams p, uint32_t chunk_count) {
  const uint32_t lane = lane_id();
  const uint32_t wave = (blockIdx.x * kBlockThreads + threadIdx.x) >> 6;
  const uint32_t wave_stride = (gridDim.x * kBlockThreads) >> 6;
  bool bad = false;
  for (uint32_t chunk = wave; chunk < chunk_count; chunk += wave_stride) {
    const uint32_t chunk_base = chunk * kAtomicChunk;
    LaneWork w = test_four(p, chunk_base, lane, bad);
    uint64_t pend[kAtomicPerLane];
#pragma unroll
    for (uint32_t k = 0; k < kAtomicPerLane; ++k) pend[k] = __ballot(w.vis[k]);
    // one iteration per distinct mesh among this wave's visible instances (usually one)
    while (any_pending(pend)) {
      uint32_t m = 0;
#pragma unroll
      for (int k = kAtomicPerLane - 1; k >= 0; --k)
        if (pend[k] != 0ull) m = __builtin_amdgcn_readlane(w.mesh[k], (int)__builtin_ctzll(pend[k]));
      uint64_t same[kAtomicPerLane];
      uint32_t total = 0;
#pragma unroll
      for (uint32_t k = 0; k < kAtomicPerLane; ++k) {
        same[k] = __ballot(w.vis[k] && w.mesh[k] == m);
        total += (uint32_t)__popcll(same[k]);
        pend[k] &= ~same[k];
      }
      uint32_t base = 0;
      if (lane == 0u) base = atomicAdd(&p.cmd_out[m].instanceCount, total);
      base = __builtin_amdgcn_readfirstlane(base) + p.mesh_base[m];
#pragma unroll
      for (uint32_t k = 0; k < kAtomicPerLane; ++k) {
        if (w.vis[k] && w.mesh[k] == m) {
          const uint32_t slot = base + rank_in(same[k]);
          if (slot < p.instance_count)
            p.visible[slot] = p.first_instance + chunk_base + k * 64u + lane;
          else
            bad = true;
        }
        base += (uint32_t)__popcll(same[k]);
      }
    }
  }
  if (bad && p.error_flag) *p.error_flag = 1u;
}

// ------------------------------------------------------------------ deterministic mode
struct DetWorkspace {
  uint64_t* mask;         // ceil(N/1024)*16 words
  uint32_t* block_count;  // nblk
  uint32_t* block_prefix; // nblk + 1
  uint32_t* block_mesh;   // nblk + 1: local mesh index of each block's first instance; [nblk] = M-1
  uint32_t* block_lo;     // nblk: mesh_base[block_mesh[b]] clamped to the block's first instance
  uint32_t* mesh_rank;    // M + 1
  uint16_t* word_prefix;  // nblk * 16: visible instances of the block before each 64-instance word
};

__host__ __device__ constexpr uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

__host__ __device__ inline DetWorkspace carve(void* base, uint64_t n, uint32_t m) {
  const uint64_t nblk = (n + kBlockInstances - 1) / kBlockInstances;
  uint8_t* b = static_cast<uint8_t*>(base);
  DetWorkspace w;
  w.mask = reinterpret_cast<uint64_t*>(b);
  b += align_up(nblk * kWordsPerBlock * 8, 256);
  w.block_count = reinterpret_cast<uint32_t*>(b);
  b += align_up(nblk * 4, 256);
  w.block_prefix = reinterpret_cast<uint32_t*>(b);
  b += align_up((nblk + 1) * 4, 256);
  w.block_mesh = reinterpret_cast<uint32_t*>(b);
  b += align_up((nblk + 1) * 4, 256);
  w.block_lo = reinterpret_cast<uint32_t*>(b);
  b += align_up(nblk * 4, 256);
  w.mesh_rank = reinterpret_cast<uint32_t*>(b);
  b += align_up(((uint64_t)m + 1) * 4, 256);
  w.word_prefix = reinterpret_cast<uint16_t*>(b);
  return w;
}

// Largest m with mesh_base[m] <= i (0 if there is none).
__device__ __forceinline__ uint32_t mesh_holding(const CullParams& p, uint32_t i) {
  uint32_t lo = 0, hi = p.mesh_count - 1u;
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1u) >> 1;
    if (p.mesh_base[mid] <= i) lo = mid; else hi = mid - 1u;
  }
  return lo;
}

// Pass A: one block per 1024 instances.
__global__ __launch_bounds__(256) void cull_mask_kernel(const CullParams p) {
  __shared__ uint32_t word_count[kWordsPerBlock];
  const DetWorkspace ws = carve(p.workspace, p.instance_count, p.mesh_count);
  const uint32_t lane = lane_id();
  const uint32_t wv = threadIdx.x >> 6;
  const uint32_t blk = blockIdx.x;
  bool bad = false;
  // rows of this block: k*256 + wv*64 + lane  (wave wv owns words k*4 + wv)
  float4 mn[kPerLane], mx[kPerLane];
  bool in[kPerLane];
#pragma unroll
  for (uint32_t k = 0; k < kPerLane; ++k) {
    const uint32_t i = blk * kBlockInstances + k * 256u + threadIdx.x;
    in[k] = i < p.instance_count;
    const uint32_t ic = in[k] ? i : 0u;
    mn[k] = load_stream(p.aabb_min + ic);
    mx[k] = load_stream(p.aabb_max + ic);
  }
  // The mesh id of every instance must agree with the segment table (deterministic mode relies on
  // contiguity).  Usually a row of 64 instances lies in one mesh: its range comes from two scalar loads,
  // and the loads of all four rows are requested together, ahead of the plane tests.  (Deriving the
  // expected mesh from the table instead - a binary search per wave while the boxes are in flight, then
  // one compare per lane - measured 1.3 us slower per call: ten dependent scalar loads outlast the boxes.)
  uint32_t mesh[kPerLane], row_lo[kPerLane], row_hi[kPerLane];
  bool row_known[kPerLane];
#pragma unroll
  for (uint32_t k = 0; k < kPerLane; ++k) {
    mesh[k] = __float_as_uint(mn[k].w) - p.first_mesh;
    const uint32_t mesh0 = __builtin_amdgcn_readfirstlane(mesh[k]);  // lane 0 is in range whenever any lane of the row is
    row_known[k] = mesh0 < p.mesh_count;
    const uint32_t mc = row_known[k] ? mesh0 : 0u;
    const uint32_t next = (mc + 1u < p.mesh_count) ? mc + 1u : mc;
    row_lo[k] = p.mesh_base[mc];
    const uint32_t after = p.mesh_base[next];
    row_hi[k] = (mc + 1u < p.mesh_count) ? after : p.instance_count;
  }
#pragma unroll
  for (uint32_t k = 0; k < kPerLane; ++k) {
    const uint32_t i = blk * kBlockInstances + k * 256u + threadIdx.x;
    bool v = in[k] && aabb_visible(mn[k], mx[k], p.planes);
    const uint64_t in_row = __ballot(in[k]);
    const uint32_t mesh0 = __builtin_amdgcn_readfirstlane(mesh[k]);
    bool ok = true;
    if (__ballot(in[k] && mesh[k] != mesh0) == 0ull) {
      const uint32_t row_first = i - lane, row_last = row_first + (uint32_t)__popcll(in_row) - 1u;
      ok = (in_row == 0ull) || (row_known[k] && row_first >= row_lo[k] && row_last < row_hi[k]);
    } else if (in[k]) {
      ok = mesh[k] < p.mesh_count;
      if (ok) {
        const uint32_t lo = p.mesh_base[mesh[k]];
        const uint32_t hi = (mesh[k] + 1u < p.mesh_count) ? p.mesh_base[mesh[k] + 1u] : p.instance_count;
        ok = (i >= lo) && (i < hi);
      }
    }
    if (in[k] && !ok) {
      bad = true;
      v = false;
    }
    const uint64_t word = __ballot(v);
    if (lane == 0u) {
      ws.mask[(uint64_t)blk * kWordsPerBlock + k * 4u + wv] = word;
      word_count[k * 4u + wv] = (uint32_t)__popcll(word);
    }
    // mesh of the block's first instance; taken from the segment table when the instance's own id is wrong,
    // so that the placement pass visits every mesh even for inconsistent input
    if (k == 0u && threadIdx.x == 0u) {
      const uint32_t m_first = ok ? mesh[k] : mesh_holding(p, i);
      ws.block_mesh[blk] = m_first;
      ws.block_lo[blk] = min(p.mesh_base[m_first], i);
    }
  }
  __syncthreads();
  if (threadIdx.x < kWordsPerBlock) {
    // exclusive prefix over the 16 word counts, in the first 16 lanes of wave 0
    const uint32_t mine = word_count[threadIdx.x];
    uint32_t s = mine;
#pragma unroll
    for (uint32_t d = 1; d < kWordsPerBlock; d <<= 1) {
      const uint32_t o = __shfl_up(s, d);
      if (threadIdx.x >= d) s += o;
    }
    ws.word_prefix[(uint64_t)blk * kWordsPerBlock + threadIdx.x] = (uint16_t)(s - mine);
    if (threadIdx.x == kWordsPerBlock - 1u) ws.block_count[blk] = s;
  }
  if (threadIdx.x == 0u && blk == gridDim.x - 1u) ws.block_mesh[gridDim.x] = p.mesh_count - 1u;
  if (bad && p.error_flag) *p.error_flag = 1u;
}

// Number of visible instances of i's block with index < i: two independent loads.
__device__ __forceinline__ uint32_t rank_in_block(const DetWorkspace& ws, uint32_t i) {
  const uint32_t blk = i / kBlockInstances;
  const uint32_t w = (i % kBlockInstances) >> 6;  // instance order inside a block is row-major over (k, wave, lane)
  const uint32_t bit = i & 63u;
  const uint64_t word = ws.mask[(uint64_t)blk * kWordsPerBlock + w];
  const uint32_t below = bit ? (uint32_t)__popcll(word & ((1ull << bit) - 1ull)) : 0u;
  return ws.word_prefix[(uint64_t)blk * kWordsPerBlock + w] + below;
}

// Global rank (number of visible instances with index < i), from the scanned block prefixes.
__device__ __forceinline__ uint32_t rank_before(const DetWorkspace& ws, uint32_t i) {
  return ws.block_prefix[i / kBlockInstances] + rank_in_block(ws, i);
}

// Pass C (same launch as B): rank of every mesh boundary, then the commands.
__device__ __forceinline__ void mesh_rank_and_cmd(const CullParams& p, const DetWorkspace& ws, uint32_t nblk, uint32_t m) {
  const uint32_t lo = p.mesh_base[m];
  const uint32_t hi = (m + 1u < p.mesh_count) ? p.mesh_base[m + 1u] : p.instance_count;
  const uint32_t r_lo = (lo < p.instance_count) ? rank_before(ws, lo) : ws.block_prefix[nblk];
  const uint32_t r_hi = (hi < p.instance_count) ? rank_before(ws, hi) : ws.block_prefix[nblk];
  ws.mesh_rank[m] = r_lo;
  if (m == p.mesh_count - 1u) ws.mesh_rank[p.mesh_count] = r_hi;
  DrawCmd c = p.cmd_template[m];
  c.instanceCount = r_hi - r_lo;
  c.firstInstance = p.first_instance + lo;
  p.cmd_out[m] = c;
}

// Pass B: exclusive scan of the block counts by one 1024-thread block, four counts per thread and pass
// (a single pass covers 2^22 instances), followed by pass C.
__global__ __launch_bounds__(1024) void cull_scan_kernel(const CullParams p, uint32_t nblk) {
  __shared__ uint32_t wave_sum[16];
  __shared__ uint32_t carry_s;
  const DetWorkspace ws = carve(p.workspace, p.instance_count, p.mesh_count);
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  if (threadIdx.x == 0u) carry_s = 0u;
  __syncthreads();
  for (uint32_t base = 0; base < nblk; base += 4096u) {
    const uint32_t i = base + threadIdx.x * 4u;
    uint32_t v[4];
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) v[k] = (i + k < nblk) ? ws.block_count[i + k] : 0u;
    const uint32_t mine = (v[0] + v[1]) + (v[2] + v[3]);
    uint32_t s = mine;  // inclusive wave scan
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
      const uint32_t o = __shfl_up(s, d);
      if (lane >= d) s += o;
    }
    if (lane == 63u) wave_sum[wv] = s;
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t k = 0; k < wv; ++k) woff += wave_sum[k];
    const uint32_t carry = carry_s;
    uint32_t run = carry + woff + s - mine;
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {
      if (i + k < nblk) ws.block_prefix[i + k] = run;
      run += v[k];
    }
    __syncthreads();
    if (threadIdx.x == 1023u) carry_s = run;
    __syncthreads();
  }
  if (threadIdx.x == 0u) ws.block_prefix[nblk] = carry_s;
  // block_prefix was written by this block: a barrier makes it visible to all of its threads
  __syncthreads();
  for (uint32_t m = threadIdx.x; m < p.mesh_count; m += 1024u) mesh_rank_and_cmd(p, ws, nblk, m);
}

// Pass D: scatter.  visible[mesh_base[m] + (g - mesh_rank[m])] = i, ascending in i.
__global__ __launch_bounds__(256) void cull_scatter_kernel(const CullParams p) {
  const DetWorkspace ws = carve(p.workspace, p.instance_count, p.mesh_count);
  const uint32_t blk = blockIdx.x;
  const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
  if (ws.block_count[blk] == 0u) return;  // nothing visible in this block
  const uint32_t block_before = ws.block_prefix[blk];
  const uint32_t m_first = ws.block_mesh[blk];
  const uint32_t m_last = ws.block_mesh[blk + 1u];
#pragma unroll
  for (uint32_t k = 0; k < kPerLane; ++k) {
    const uint32_t w = k * 4u + wv;
    const uint64_t word = ws.mask[(uint64_t)blk * kWordsPerBlock + w];
    if ((word >> lane) & 1ull) {
      const uint32_t i = blk * kBlockInstances + w * 64u + lane;
      const uint32_t g = block_before + ws.word_prefix[(uint64_t)blk * kWordsPerBlock + w] + rank_in(word);
      // largest m in [m_first, m_last] with mesh_base[m] <= i
      uint32_t lo = m_first, hi = m_last;
      while (lo < hi) {
        const uint32_t mid = (lo + hi + 1u) >> 1;
        if (p.mesh_base[mid] <= i) lo = mid; else hi = mid - 1u;
      }
      const uint32_t slot = p.mesh_base[lo] + (g - ws.mesh_rank[lo]);
      if (slot < p.instance_count) p.visible[slot] = p.first_instance + i;
    }
  }
}

// Passes B-D in one launch, for up to kPlaceMaxBlocks workgroups.  Workgroup b needs two prefix sums of
// the block counts: up to itself, and up to the block in which the mesh of its first instance begins (the
// only mesh of this workgroup whose first slot lies in an earlier block).  Both come out of one coalesced
// pass over block_count[0, b), which stays in L2: 16 KiB at 2^22 instances.
// The command of mesh m is written by the workgroup whose instance range (B0, B1] holds the mesh's end;
// that workgroup knows the rank at both ends of the mesh.
// The pass is bound by the latency of dependent loads, not by bytes: everything whose address depends only
// on the workgroup index is requested first, which leaves the usual workgroup (one mesh throughout) with a
// single dependent level, the rank at which that mesh begins.
constexpr uint32_t kPlaceMaxBlocks = 8192;
constexpr uint32_t kPlaceSumTrips = kPlaceMaxBlocks / (kBlockThreads * 4u);

__global__ __launch_bounds__(256) void cull_place_kernel(const CullParams p, uint32_t nblk) {
  __shared__ uint32_t s_part[2][kBlockThreads / 64];
  const DetWorkspace ws = carve(p.workspace, p.instance_count, p.mesh_count);
  const uint32_t blk = blockIdx.x;
  const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
  const uint32_t n = p.instance_count;
  const uint32_t b0 = blk * kBlockInstances, b1 = min(b0 + kBlockInstances, n);

  // level 1: addresses known from the workgroup index alone
  const uint32_t m_first = ws.block_mesh[blk];
  const uint32_t m_last = ws.block_mesh[blk + 1u];
  const uint32_t lo_first = ws.block_lo[blk];  // <= b0
  const uint32_t count_here = ws.block_count[blk];
  uint64_t word[kPerLane];
  uint32_t word_before[kPerLane];
#pragma unroll
  for (uint32_t k = 0; k < kPerLane; ++k) {
    word[k] = ws.mask[(uint64_t)blk * kWordsPerBlock + k * 4u + wv];
    word_before[k] = ws.word_prefix[(uint64_t)blk * kWordsPerBlock + k * 4u + wv];
  }
  uint4 counts[kPlaceSumTrips];
#pragma unroll
  for (uint32_t it = 0; it < kPlaceSumTrips; ++it) {
    const uint32_t j = it * kBlockThreads * 4u + threadIdx.x * 4u;
    counts[it] = make_uint4(0u, 0u, 0u, 0u);
    if (j < blk) counts[it] = *reinterpret_cast<const uint4*>(ws.block_count + j);  // the array is padded to 256 bytes
  }
  // level 2: rank inside its block of the instance at which mesh m_first begins
  const uint32_t blk_first = lo_first / kBlockInstances;
  const uint32_t rank_first_in_block = rank_in_block(ws, lo_first);

  uint32_t sum_first = 0, sum_here = 0;
#pragma unroll
  for (uint32_t it = 0; it < kPlaceSumTrips; ++it) {
    const uint32_t j = it * kBlockThreads * 4u + threadIdx.x * 4u;
    const uint32_t v[4] = {counts[it].x, counts[it].y, counts[it].z, counts[it].w};
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {
      if (j + k < blk) sum_here += v[k];
      if (j + k < blk_first) sum_first += v[k];
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    sum_first += __shfl_xor(sum_first, off);
    sum_here += __shfl_xor(sum_here, off);
  }
  if (lane == 0u) {
    s_part[0][wv] = sum_first;
    s_part[1][wv] = sum_here;
  }
  __syncthreads();
  sum_first = (s_part[0][0] + s_part[0][1]) + (s_part[0][2] + s_part[0][3]);
  sum_here = (s_part[1][0] + s_part[1][1]) + (s_part[1][2] + s_part[1][3]);
  const uint32_t rank_first = sum_first + rank_first_in_block;  // rank at which mesh m_first begins

  // rank of boundary i, b0 <= i <= b1
  auto rank_at = [&](uint32_t i) -> uint32_t { return sum_here + ((i >= b0 + kBlockInstances) ? count_here : rank_in_block(ws, i)); };

  // commands of the meshes that end inside (b0, b1]; none when one mesh covers the block and goes on
  const bool ends_here = (m_first != m_last) || (blk == 0u) || (b1 == n);
  if (ends_here) {
    for (uint32_t m = (blk == 0u ? 0u : m_first) + threadIdx.x; m <= m_last; m += kBlockThreads) {
      const uint32_t lo = p.mesh_base[m];
      const uint32_t hi = (m + 1u < p.mesh_count) ? p.mesh_base[m + 1u] : n;
      const uint32_t lo_c = min(lo, n), hi_c = min(hi, n);
      if (!((hi_c > b0 && hi_c <= b1) || (hi_c == 0u && blk == 0u))) continue;
      const uint32_t r_hi = (hi_c == 0u) ? 0u : rank_at(hi_c);
      const uint32_t r_lo = (hi_c == 0u) ? 0u : (lo_c < b0 ? rank_first : rank_at(min(lo_c, hi_c)));
      DrawCmd c = p.cmd_template[m];
      c.instanceCount = r_hi - r_lo;
      c.firstInstance = p.first_instance + lo;
      p.cmd_out[m] = c;
    }
  }

  if (count_here == 0u) return;  // nothing visible in this block
#pragma unroll
  for (uint32_t k = 0; k < kPerLane; ++k) {
    if ((word[k] >> lane) & 1ull) {
      const uint32_t i = b0 + (k * 4u + wv) * 64u + lane;
      const uint32_t g = sum_here + word_before[k] + rank_in(word[k]);
      uint32_t base = lo_first, r = rank_first;
      if (m_first != m_last) {
        // largest m in [m_first, m_last] with mesh_base[m] <= i
        uint32_t lo = m_first, hi = m_last;
        while (lo < hi) {
          const uint32_t mid = (lo + hi + 1u) >> 1;
          if (p.mesh_base[mid] <= i) lo = mid; else hi = mid - 1u;
        }
        if (lo != m_first) {
          base = p.mesh_base[lo];
          r = rank_at(min(max(base, b0), b1));
        }
      }
      const uint32_t slot = base + (g - r);
      if (slot < n) p.visible[slot] = p.first_instance + i;
    }
  }
}

constexpr uint64_t workspace_size(uint64_t n, uint32_t m) {
  const uint64_t nblk = (n + kBlockInstances - 1) / kBlockInstances;
  return align_up(nblk * kWordsPerBlock * 8, 256) + 2 * align_up(nblk * 4, 256) + 2 * align_up((nblk + 1) * 4, 256) +
         align_up(((uint64_t)m + 1) * 4, 256) + align_up(nblk * kWordsPerBlock * 2, 256);
}

}  // namespace

// Workspace of the deterministic mode for callers that must not allocate: one array per GPU, part of the
// code object, sized for the largest call the allocation-free entry point accepts (every term of
// workspace_size grows with n and m, so any smaller call fits).  Host code never passes `scratch` itself to
// a kernel or a memory call - on the host that name is only a handle for hipGetSymbolAddress; the device
// address it yields differs per GPU and is kept in g_scratch_ptr.
constexpr uint64_t kScratchBytes = workspace_size(kCullStaticMaxInstances, kCullStaticMaxMeshes);
static_assert(kScratchBytes <= (1ull << 20), "the static workspace is meant to stay modest");
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
void* g_scratch_ptr[kMaxDevices] = {};
}  // namespace

uint64_t vkr_cull_workspace_size(uint64_t n, uint32_t m) { return workspace_size(n, m); }

hipError_t vkr_cull_static_init(int device) {
  if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (g_scratch_ptr[device]) return hipSuccess;
  void* ptr = nullptr;
  const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));  // the address on the current device
  if (e != hipSuccess) return e;
  if (!ptr || (reinterpret_cast<uintptr_t>(ptr) & 255u)) return hipErrorInvalidSymbol;
  g_scratch_ptr[device] = ptr;
  return hipSuccess;
}

void* vkr_cull_static_workspace(int device, uint64_t* out_bytes) {
  if (out_bytes) *out_bytes = kScratchBytes;
  if (device < 0 || device >= kMaxDevices) return nullptr;
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  return g_scratch_ptr[device];
}

hipError_t vkr_launch_cull_atomic(const CullParams& p, hipStream_t stream) {
  if (p.mesh_count == 0u) return hipSuccess;
  const uint32_t chunks = (p.instance_count + kAtomicChunk - 1u) / kAtomicChunk;
  // 4 waves per block; cap the grid at 8 blocks per CU and let waves stride over chunks
  uint32_t blocks = (chunks + 3u) / 4u;
  if (blocks > 2048u) blocks = 2048u;
  hipLaunchKernelGGL(cull_init_cmds, dim3((p.mesh_count + 255u) / 256u), dim3(256), 0, stream, p);
  if (p.instance_count == 0u) return hipGetLastError();
  hipLaunchKernelGGL(cull_atomic_kernel, dim3(blocks), dim3(kBlockThreads), 0, stream, p, chunks);
  return hipGetLastError();
}

hipError_t vkr_launch_cull_deterministic(const CullParams& p, hipStream_t stream) {
  if (p.mesh_count == 0u) return hipSuccess;
  const uint32_t nblk = (p.instance_count + kBlockInstances - 1u) / kBlockInstances;
  if (nblk == 0u) {
    hipLaunchKernelGGL(cull_init_cmds, dim3((p.mesh_count + 255u) / 256u), dim3(256), 0, stream, p);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(cull_mask_kernel, dim3(nblk), dim3(kBlockThreads), 0, stream, p);
  if (nblk <= kPlaceMaxBlocks) {
    hipLaunchKernelGGL(cull_place_kernel, dim3(nblk), dim3(kBlockThreads), 0, stream, p, nblk);
  } else {
    hipLaunchKernelGGL(cull_scan_kernel, dim3(1), dim3(1024), 0, stream, p, nblk);
    hipLaunchKernelGGL(cull_scatter_kernel, dim3(nblk), dim3(kBlockThreads), 0, stream, p);
  }
  return hipGetLastError();
}
