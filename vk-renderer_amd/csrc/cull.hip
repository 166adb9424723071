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
//   deterministic mode : one launch; a resident grid walks the blocks of 1024 instances, each block
//                        publishes its visible count in a status word and looks back through its
//                        predecessors' words for the rank at which its first mesh continues, then
//                        places its indices in ascending order and writes the commands of the
//                        meshes that end inside it.

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

// Visibility and local mesh index (global id - first_mesh) of this lane's instances.  Instance of (wave chunk,
// row k, lane l) is chunk*256 + k*64 + l, so every wave-level load covers 1 KiB of contiguous memory.
struct LaneWork {
  bool vis[kAtomicPerLane];
  uint32_t mesh[kAtomicPerLane];
};

// Counts accumulate in cmd_out[m].instanceCount, zeroed by cull_init_cmds in the launch before.
// Folding both into one launch was tried twice and is no faster than the 4.5 us the extra launch costs:
// (a) the first blocks initialise the records and bump a counter that every wave polls before its
// first atomic - the waves reach the poll before the initialising blocks have pushed their stores out
// (__threadfence) and stall for about as long as the launch took; with an agent-scope *acquire* in
// the poll it is 3x slower, because that invalidates the XCD's L2 once per wave (the L2s of the
// eight XCDs are not coherent with each other);  (b) a "last block" ticket at the end was 4x slower;
// (c) no initialisation at all: a 64-bit word per mesh in context-owned memory (tested << 32 | visible, zeroed
// by the call before) takes every wave's count with one atomic, and the wave whose addition completes the
// tested half writes the mesh's command - correct (every test passed), but every wave now needs an atomic,
// not only those with a visible instance, and the call took 32.9 us (31.6 without checking the ids against
// the segment table) against 27.5 with the two launches.
//
// A wave owns chunks of 256 consecutive instances.  Memory operations of a wave return in the order they were
// issued, so the order of the requests decides what a wait covers: the atomic that reserves the wave's slots
// goes out first, then the load of the segment base, then the boxes of the wave's NEXT chunk, and the wait for
// the atomic's result leaves those eight loads in flight (s_waitcnt vmcnt(8)).  30.5 -> 27.4 us per call at
// 2^22 instances against a loop that asked for each chunk's boxes at the top of its trip.  What that takes:
//   * the compiler's wave-level atomic optimiser is off for this file (-amdgpu-atomic-optimizer-strategy=None):
//     it would wrap the one-lane atomic in a reduction of its own and wait for the result straight away;
//   * the segment base is asked for through a vector register and made uniform only after the requests, else
//     the value is moved to a scalar register, and waited for, where it is requested;
//   * the requests are unconditional (a wave without a next chunk asks for instance 0): a wait that follows a
//     branch with and a branch without requests is sized for the branch without;
//   * the unused fourth component of the max corners counts as read when a box is tested: a register that is
//     free while its load is in flight is handed to a temporary, whose write then has to wait for that load.
// Measured alternatives (us per call): the requests ahead of the atomic, so that the atomic's result comes back
// behind them: 31-32; two load groups of 256 instances per atomic: 32 (four: 40); 128 instances per chunk: 30.5;
// 512: 30.7; grids of 1024 / 1280 / 1536 / 1792 / 2048 workgroups: 28.4 / 28.2 / 27.4 / 28.4 / 27.5.
__global__ __launch_bounds__(256) void cull_atomic_kernel(const CullParams p, uint32_t chunk_count) {
  const uint32_t lane = lane_id();
  const uint32_t wave = (blockIdx.x * kBlockThreads + threadIdx.x) >> 6;
  const uint32_t wave_stride = (gridDim.x * kBlockThreads) >> 6;
  const uint32_t n = p.instance_count;
  bool bad = false;
  float4 mn[kAtomicPerLane], mx[kAtomicPerLane];
  auto request_boxes = [&](uint32_t chunk) {
#pragma unroll
    for (uint32_t k = 0; k < kAtomicPerLane; ++k) {
      const uint32_t i = chunk * kAtomicChunk + k * 64u + lane;
      const uint32_t ic = i < n ? i : 0u;
      mn[k] = load_stream(p.aabb_min + ic);
      mx[k] = load_stream(p.aabb_max + ic);
    }
  };
  if (wave < chunk_count) request_boxes(wave);
#pragma unroll 1
  for (uint32_t chunk = wave; chunk < chunk_count; chunk += wave_stride) {
    const uint32_t chunk_base = chunk * kAtomicChunk;
    LaneWork w;
#pragma unroll
    for (uint32_t k = 0; k < kAtomicPerLane; ++k) {
      const uint32_t i = chunk_base + k * 64u + lane;
      w.mesh[k] = __float_as_uint(mn[k].w) - p.first_mesh;
      // the unused fourth component counts as read here: a register that is free while its load is still in
      // flight would be handed to a temporary, whose write has to wait for that load
      asm volatile("" : : "v"(mx[k].w));
      bool v = i < n && aabb_visible(mn[k], mx[k], p.planes);
      if (v && w.mesh[k] >= p.mesh_count) {
        bad = true;
        v = false;
      }
      w.vis[k] = v;
    }
    uint64_t pend[kAtomicPerLane];
#pragma unroll
    for (uint32_t k = 0; k < kAtomicPerLane; ++k) pend[k] = __ballot(w.vis[k]);
    bool next_requested = false;
    const uint32_t next = chunk + wave_stride;
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
      if (lane == 0u) base = __hip_atomic_fetch_add(&p.cmd_out[m].instanceCount, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // where the mesh's segment begins: asked for through a vector register, so that the value is not moved to a
      // scalar register (and waited for) the moment it is requested
      uint32_t m_lane;
      asm volatile("v_mov_b32 %0, %1" : "=v"(m_lane) : "s"(m));
      uint32_t segment = p.mesh_base[m_lane];
      if (!next_requested) {
        asm volatile("" ::: "memory");  // the loads below stay behind the atomic
        request_boxes(next < chunk_count ? next : 0xFFFFFFFFu / kAtomicChunk);  // beyond the end: every lane reads instance 0
        next_requested = true;
      }
      asm volatile("" : "+v"(base), "+v"(segment) : : "memory");  // first use of the two results: after the requests
      base = __builtin_amdgcn_readfirstlane(base) + __builtin_amdgcn_readfirstlane(segment);
#pragma unroll
      for (uint32_t k = 0; k < kAtomicPerLane; ++k) {
        if (w.vis[k] && w.mesh[k] == m) {
          const uint32_t slot = base + rank_in(same[k]);
          if (slot < n)
            p.visible[slot] = p.first_instance + chunk_base + k * 64u + lane;
          else
            bad = true;
        }
        base += (uint32_t)__popcll(same[k]);
      }
    }
    if (!next_requested && next < chunk_count) request_boxes(next);
  }
  if (bad && p.error_flag) *p.error_flag = 1u;
}

// ------------------------------------------------------------------ deterministic mode
// Largest m with mesh_base[m] <= i (0 if there is none).
__device__ __forceinline__ uint32_t mesh_holding(const CullParams& p, uint32_t i) {
  uint32_t lo = 0, hi = p.mesh_count - 1u;
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1u) >> 1;
    if (p.mesh_base[mid] <= i) lo = mid; else hi = mid - 1u;
  }
  return lo;
}

// A compacted list is ordered inside each mesh segment only, and the segment bases are known beforehand
// (mesh_base), so the rank a workgroup needs from its predecessors is the number of visible instances of ITS
// FIRST MESH that lie before it - a sum that reaches back to the workgroup in which that mesh begins and no
// further (typically 0-4 workgroups), not a prefix over the whole scene.  Workgroups publish one 64-bit status
// word each and look back through their predecessors' words:
//   PREFIX    value = visible instances of the workgroup's LAST mesh before the workgroup's end.  Published
//             at once when that mesh begins inside the workgroup (nothing earlier contributes), else after
//             the look-back;
//   AGGREGATE value = visible instances of the workgroup, published at once by a workgroup that lies inside
//             one mesh begun earlier, so that successors can walk past it while it is still looking back.
// The words carry the call's sequence number, so a word left by an earlier call reads as "not yet published"
// and nothing has to be cleared between calls.  Status words are written and polled with relaxed agent-scope
// atomics (the value is the message: no other memory has to become visible between workgroups).
// Why the waits end.  A block waits only for lower-numbered blocks, and every workgroup publishes its block's
// first word before it waits for anything, so a wait can only be for a block whose workgroup has not got that far
// yet.  With the device to itself the whole grid is resident (it is sized for that) and that is a matter of
// microseconds.  When other work holds part of the device - another context's kernel, another stream - a
// workgroup of this grid may still be waiting for a place while later-numbered ones (dispatched to a compute die
// with room) already poll its word: it gets its place when a workgroup of the other kernel retires, so the wait is
// as long as the other kernel's workgroups take, and ends.  The wait is therefore bounded by the wall clock, not by
// a number of polls: only after kCullPatienceSeconds without the awaited word - the other workgroups cannot run at
// all: a device fault, or foreign work that itself never ends - does a workgroup stop waiting, so that the GPU is
// not held for ever.  The launch's results are then wrong, and it says so twice: through error_flag (if given) and
// through the context's fault word in host memory, which every synchronising call of the library checks and
// reports as VKR_ERR_DEVICE.
// The command of mesh m is written by the block in which the mesh ends.
// Measured at 2^22 instances, 1024 meshes: 32.5 us per call, against 37.5 us for the former mask launch +
// placement launch.  One workgroup per block with the look-back straight after the tests measured 39-41 us (the
// block idles through the round trips of its polls while holding its wave slots), the same with block numbers
// drawn from a ticket counter 75 us (4096 returning atomics on one address serialise).
// The launch is sensitive to how many workgroups are resident: capped at 512 / 768 / 1024 / 1280 of the 1536 that
// 75 vector registers allow, a call takes 55 / 43.5 / 38.6 / 34.9 us against 32.4.  Getting to seven or eight
// waves per SIMD was tried through the launch bound (72 registers: 5 spilled, 37.7 us; 64: 16 spilled, 43 us),
// through 12-byte loads of the max corners (no fewer registers, the loads slower: 33.7 us) and by keeping the
// per-lane segment bases in LDS (73 registers, one too many); scalar registers are the tighter resource (56
// spilled into vector lanes, the six planes taking 24): re-reading the planes from the argument block per trip
// made the compiler copy the block to scratch memory, and fetching them from LDS per trip raised the count to 79.

constexpr uint32_t kKindAggregate = 1u, kKindPrefix = 2u;

__device__ __forceinline__ void publish_status(uint64_t* slot, uint32_t seq, uint32_t kind, uint32_t value) {
  __hip_atomic_store(slot, ((uint64_t)((seq << 2) | kind) << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A workgroup of the resident grid takes the blocks blockIdx.x, blockIdx.x + gridDim.x, ... of 1024 instances
// (rows of a block: k*256 + wv*64 + lane; wave wv owns words k*4 + wv) and works on two of them at a time:
//   phase A of block i   : plane tests, visibility words into LDS, first status word published;
//   phase F of block i-1 : look-back, commands, placement - one trip later, when the predecessors of block i-1
//                          published long ago and the first poll finds them;
// and the boxes of block i+1 are requested between the two, so loads are in flight throughout.
struct BlockMeta {  // of the block's first / last instance: mesh, where its segment begins (and ends)
  uint32_t m_first, lo_first, m_last, lo_last, hi_last;
};

__global__ __launch_bounds__(256) void cull_onepass_kernel(const CullParams p, const OnePassState o, uint32_t nblk) {
  __shared__ uint64_t s_word[2][kWordsPerBlock];
  __shared__ uint32_t s_before[2][kWordsPerBlock + 1];  // visible instances of the block before each word; [16] = all
  __shared__ BlockMeta s_meta[2];
  __shared__ uint32_t s_rank_first;
  const uint32_t lane = lane_id();
  const uint32_t wv = threadIdx.x >> 6;
  const uint32_t n = p.instance_count;
  bool bad = false;
  float4 mn[kPerLane], mx[kPerLane];
  auto request_boxes = [&](uint32_t block) {
#pragma unroll
    for (uint32_t k = 0; k < kPerLane; ++k) {
      const uint32_t i = block * kBlockInstances + k * 256u + threadIdx.x;
      const uint32_t ic = i < n ? i : 0u;
      mn[k] = load_stream(p.aabb_min + ic);
      mx[k] = load_stream(p.aabb_max + ic);
    }
  };

  // Phase F for block `blk` whose words are in LDS buffer `buf`; seg_lo: where the segment of each of this lane's
  // four instances begins.  `first_poll` is wave 0's status word of its nearest predecessors, requested earlier.
  auto rank_in_block = [&](uint32_t buf, uint32_t b0, uint32_t i) -> uint32_t {  // visible instances of the block before i, b0 <= i <= b1
    const uint32_t w = (i - b0) >> 6;
    if (w >= kWordsPerBlock) return s_before[buf][kWordsPerBlock];
    const uint32_t bit = i & 63u;  // b0 is a multiple of 64
    return s_before[buf][w] + (bit ? (uint32_t)__popcll(s_word[buf][w] & ((1ull << bit) - 1ull)) : 0u);
  };
  auto poll = [&](int64_t j) -> uint64_t { return __hip_atomic_load(o.status + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto look_back = [&](uint32_t blk, uint32_t buf, uint64_t first_poll) {  // wave 0 only
    const uint32_t b0 = blk * kBlockInstances;
    const BlockMeta meta = s_meta[buf];
    uint32_t rank_first = 0;
    if (meta.lo_first < b0) {  // the first mesh began in an earlier block (so blk > 0): 64 predecessors per step, nearest in lane 0
      bool found = false;
      bool have_first = true;
      for (int64_t top = (int64_t)blk - 1; top >= 0 && !found; top -= 64) {
        const int64_t j = top - (int64_t)lane;
        uint64_t v = 0;
        if (j >= 0) {
          uint32_t polls = 0;
          uint64_t since = 0;  // wall clock at the 1024th poll in vain
          v = have_first ? first_poll : poll(j);
          while ((uint32_t)(v >> 34) != o.seq) {
            if ((++polls & 1023u) == 0u) {
              const uint64_t now = (uint64_t)wall_clock64();
              if (since == 0ull) {
                since = now | 1ull;
              } else if (now - since > o.patience_ticks) {  // the awaited workgroup cannot run: stop, and say so
                bad = true;
                if (o.fault) __hip_atomic_store(o.fault, o.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                v = ((uint64_t)((o.seq << 2) | kKindPrefix) << 32);
                break;
              }
            }
            __builtin_amdgcn_s_sleep(2);
            v = poll(j);
          }
        }
        have_first = false;
        const uint32_t kind = (uint32_t)(v >> 32) & 3u;
        const uint64_t prefixes = __ballot(j >= 0 && kind == kKindPrefix);
        uint32_t add = (j >= 0) ? (uint32_t)v : 0u;
        if (prefixes != 0ull) {
          found = true;
          if (lane > (uint32_t)__builtin_ctzll(prefixes)) add = 0u;  // predecessors before the nearest PREFIX are in it
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) add += __shfl_xor(add, off);
        rank_first += add;
      }
    }
    if (lane == 0u) {
      if (meta.lo_last < b0) publish_status(o.status + blk, o.seq, kKindPrefix, rank_first + s_before[buf][kWordsPerBlock]);
      s_rank_first = rank_first;
    }
  };
  auto finish_block = [&](uint32_t blk, uint32_t buf, const uint32_t (&seg_lo)[kPerLane]) {  // after look_back and a barrier
    const uint32_t b0 = blk * kBlockInstances, b1 = min(b0 + kBlockInstances, n);
    const BlockMeta meta = s_meta[buf];
    const uint32_t rank_first = s_rank_first;
    // Commands of the meshes that end inside (b0, b1] (and, in block 0, of empty meshes at instance 0).  Most blocks
    // lie inside one mesh that goes on beyond them and have none.
    if (meta.m_first != meta.m_last || meta.hi_last <= b1 || blk == 0u) {
      for (uint32_t m0 = (blk == 0u) ? 0u : meta.m_first;; m0 += kBlockThreads) {
        const uint32_t m = m0 + threadIdx.x;
        if (m < p.mesh_count) {
          const uint32_t lo = p.mesh_base[m];
          const uint32_t hi = (m + 1u < p.mesh_count) ? p.mesh_base[m + 1u] : n;
          const uint32_t lo_c = min(lo, n), hi_c = min(hi, n);
          if ((hi_c > b0 && hi_c <= b1) || (hi_c == 0u && blk == 0u)) {
            const uint32_t r_hi = (hi_c == 0u) ? 0u : rank_in_block(buf, b0, hi_c);
            uint32_t visible_count;
            if (hi_c == 0u)
              visible_count = 0u;
            else if (lo_c < b0)
              visible_count = rank_first + r_hi;  // the mesh that holds b0
            else
              visible_count = r_hi - rank_in_block(buf, b0, min(lo_c, hi_c));
            DrawCmd c = p.cmd_template[m];
            c.instanceCount = visible_count;
            c.firstInstance = p.first_instance + lo;
            p.cmd_out[m] = c;
          }
        }
        // meshes are ordered: go on only if the last mesh of this round still ends at or before b1 (block-uniform)
        const uint32_t last = m0 + kBlockThreads - 1u;
        if (last >= p.mesh_count) break;
        const uint32_t last_hi = (last + 1u < p.mesh_count) ? p.mesh_base[last + 1u] : n;
        if (min(last_hi, n) > b1) break;
      }
    }
    if (s_before[buf][kWordsPerBlock] != 0u) {
#pragma unroll
      for (uint32_t k = 0; k < kPerLane; ++k) {
        const uint64_t word = s_word[buf][k * 4u + wv];
        if ((word >> lane) & 1ull) {
          const uint32_t i = b0 + k * 256u + threadIdx.x;
          const uint32_t g = s_before[buf][k * 4u + wv] + rank_in(word);
          const uint32_t base = seg_lo[k];
          // a segment begun before this block is the first mesh's: its ranks continue from the look-back
          const uint32_t slot = base < b0 ? base + rank_first + g : base + (g - rank_in_block(buf, b0, min(base, b1)));
          if (slot < n) p.visible[slot] = p.first_instance + i;
        }
      }
    }
  };

  request_boxes(blockIdx.x);
  uint32_t prev_blk = 0xFFFFFFFFu, prev_seg_lo[kPerLane] = {};
  uint32_t buf = 0;
#pragma unroll 1
  for (uint32_t blk = blockIdx.x; blk < nblk; blk += gridDim.x, buf ^= 1u) {
    const uint32_t b0 = blk * kBlockInstances, b1 = min(b0 + kBlockInstances, n);
    // ---- phase A.  The mesh id of every instance must agree with the segment table.  Usually a row of 64 instances
    // lies in one mesh: its range comes from two scalar loads, requested for all four rows ahead of the plane tests.
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
      row_hi[k] = (mc + 1u < p.mesh_count) ? after : n;
    }
    uint32_t seg_lo[kPerLane];
#pragma unroll
    for (uint32_t k = 0; k < kPerLane; ++k) {
      const uint32_t i = b0 + k * 256u + threadIdx.x;
      const bool in = i < n;
      bool v = in && aabb_visible(mn[k], mx[k], p.planes);
      const uint64_t in_row = __ballot(in);
      const uint32_t mesh0 = __builtin_amdgcn_readfirstlane(mesh[k]);
      bool ok = true;
      uint32_t lo = row_lo[k], hi = row_hi[k];
      if (__ballot(in && mesh[k] != mesh0) == 0ull) {
        const uint32_t row_first = i - lane, row_last = row_first + (uint32_t)__popcll(in_row) - 1u;
        ok = (in_row == 0ull) || (row_known[k] && row_first >= row_lo[k] && row_last < row_hi[k]);
      } else if (in) {
        ok = mesh[k] < p.mesh_count;
        if (ok) {
          lo = p.mesh_base[mesh[k]];
          hi = (mesh[k] + 1u < p.mesh_count) ? p.mesh_base[mesh[k] + 1u] : n;
          ok = (i >= lo) && (i < hi);
        }
      }
      if (in && !ok) {
        bad = true;
        v = false;
      }
      seg_lo[k] = lo;
      const uint64_t word = __ballot(v);
      if (lane == 0u) s_word[buf][k * 4u + wv] = word;
      // mesh and segment of the block's first and last instance; from the segment table when the instance's own id
      // is wrong, so that every mesh gets its command even for inconsistent input
      if (i == b0 || i == b1 - 1u) {
        uint32_t m = mesh[k];
        if (!ok) {
          m = mesh_holding(p, i);
          lo = p.mesh_base[m];
          hi = (m + 1u < p.mesh_count) ? p.mesh_base[m + 1u] : n;
        }
        if (i == b0) {
          s_meta[buf].m_first = m;
          s_meta[buf].lo_first = lo;
        }
        if (i == b1 - 1u) {
          s_meta[buf].m_last = m;
          s_meta[buf].lo_last = lo;
          s_meta[buf].hi_last = hi;
        }
      }
    }
    const uint32_t blk_next = blk + gridDim.x;
    // the boxes are consumed: waves 1-3 ask for the next block's at once, wave 0 once its poll is on the way
    // (memory returns in order within a wave: a poll requested after the boxes would wait for them)
    if (wv != 0u && blk_next < nblk) request_boxes(blk_next);
    __syncthreads();
    if (wv == 0u) {
      const uint32_t mine = lane < kWordsPerBlock ? (uint32_t)__popcll(s_word[buf][lane]) : 0u;
      uint32_t incl = mine;
#pragma unroll
      for (uint32_t d = 1; d < kWordsPerBlock; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
      }
      if (lane < kWordsPerBlock) s_before[buf][lane] = incl - mine;
      if (lane == kWordsPerBlock - 1u) s_before[buf][kWordsPerBlock] = incl;
      const uint32_t count = __shfl(incl, (int)(kWordsPerBlock - 1u));
      if (lane == 0u) {
        const uint32_t lo_last = s_meta[buf].lo_last;
        if (lo_last >= b0)
          publish_status(o.status + blk, o.seq, kKindPrefix, count - rank_in_block(buf, b0, min(lo_last, b1)));
        else
          publish_status(o.status + blk, o.seq, kKindAggregate, count);
      }
      // ---- phase F of the previous block, part 1 (wave 0): poll, next boxes, look back
      uint64_t first_poll = 0;
      const bool have_prev = prev_blk != 0xFFFFFFFFu;
      if (have_prev && (int64_t)prev_blk - 1 - (int64_t)lane >= 0) first_poll = poll((int64_t)prev_blk - 1 - (int64_t)lane);
      if (blk_next < nblk) request_boxes(blk_next);
      if (have_prev) look_back(prev_blk, buf ^ 1u, first_poll);
    }
    __syncthreads();
    if (prev_blk != 0xFFFFFFFFu) finish_block(prev_blk, buf ^ 1u, prev_seg_lo);
    prev_blk = blk;
#pragma unroll
    for (uint32_t k = 0; k < kPerLane; ++k) prev_seg_lo[k] = seg_lo[k];
    __syncthreads();  // the previous block's words have been read: the next phase A may overwrite them
  }
  // the workgroup's last block
  if (prev_blk != 0xFFFFFFFFu) {
    if (wv == 0u) {
      uint64_t first_poll = 0;
      if ((int64_t)prev_blk - 1 - (int64_t)lane >= 0) first_poll = poll((int64_t)prev_blk - 1 - (int64_t)lane);
      look_back(prev_blk, buf ^ 1u, first_poll);
    }
    __syncthreads();
    finish_block(prev_blk, buf ^ 1u, prev_seg_lo);
  }
  if (bad && p.error_flag) *p.error_flag = 1u;
}

}  // namespace

// State of the deterministic mode for callers that must not allocate: one array per GPU, part of the code
// object, sized for the largest call the allocation-free entry point accepts.  Host code never passes
// `scratch` itself to a kernel or a memory call - on the host that name is only a handle for
// hipGetSymbolAddress; the device address it yields differs per GPU and is kept in g_scratch[device].ptr.
// The array starts zeroed (static storage) and is cleared again only when its sequence numbers run out.
constexpr uint64_t kScratchBytes = kCullStaticMaxInstances / kBlockInstances * 8u;
static_assert(kScratchBytes == 32768, "one status word per 1024 instances");
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

namespace {
constexpr int kMaxDevices = 64;
struct StaticState {
  void* ptr = nullptr;
  hipEvent_t last_use = nullptr;  // recorded behind the most recent launch that used the array
  uint32_t seq = 0;               // sequence number of that launch
};
std::mutex g_scratch_mutex;
StaticState g_scratch[kMaxDevices];
}  // namespace

hipError_t vkr_cull_static_init(int device) {
  if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  StaticState& st = g_scratch[device];
  if (!st.ptr) {
    void* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));  // the address on the current device
    if (e != hipSuccess) return e;
    if (!ptr || (reinterpret_cast<uintptr_t>(ptr) & 255u)) return hipErrorInvalidSymbol;
    st.ptr = ptr;
  }
  if (!st.last_use) {
    const hipError_t e = hipEventCreateWithFlags(&st.last_use, hipEventDisableTiming);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

uint64_t vkr_cull_state_bytes(uint64_t instance_count) {
  const uint64_t nblk = (instance_count + kBlockInstances - 1) / kBlockInstances;
  return nblk * 8u;
}

uint32_t vkr_cull_blocks_per_cu() {
  static const uint32_t per_cu = [] {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, cull_onepass_kernel, (int)kBlockThreads, 0) != hipSuccess || nb < 1) nb = 1;
    return (uint32_t)nb;
  }();
  return per_cu;
}

namespace {
uint64_t patience_ticks() {  // kCullPatienceSeconds on the device's wall clock (100 MHz where the attribute is missing)
  static const uint64_t ticks = [] {
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
    return (uint64_t)khz * 1000ull * kCullPatienceSeconds;
  }();
  return ticks;
}
}  // namespace

hipError_t vkr_launch_cull_deterministic(const CullParams& p, void* state_mem, uint32_t seq, uint32_t resident_blocks, uint32_t* fault, hipStream_t stream) {
  if (p.mesh_count == 0u) return hipSuccess;
  const uint32_t nblk = (p.instance_count + kBlockInstances - 1u) / kBlockInstances;
  if (nblk == 0u) {
    hipLaunchKernelGGL(cull_init_cmds, dim3((p.mesh_count + 255u) / 256u), dim3(256), 0, stream, p);
    return hipGetLastError();
  }
  if (!state_mem || seq == 0u || seq >= kCullSeqEnd || resident_blocks == 0u) return hipErrorInvalidValue;
  OnePassState o;
  o.status = static_cast<uint64_t*>(state_mem);
  o.fault = fault;
  o.patience_ticks = patience_ticks();
  o.seq = seq;
  // a grid the idle device holds whole: a block's predecessors are then all at work (see the kernel's comment for
  // what happens when the device is not idle)
  hipLaunchKernelGGL(cull_onepass_kernel, dim3(min(nblk, resident_blocks)), dim3(kBlockThreads), 0, stream, p, o, nblk);
  return hipGetLastError();
}

hipError_t vkr_launch_cull_static(const CullParams& p, int device, uint32_t resident_blocks, uint32_t* fault, hipStream_t stream) {
  if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
  if (vkr_cull_state_bytes(p.instance_count) > kScratchBytes) return hipErrorInvalidValue;
  // the capture state first: a recorded launch draws no sequence number and leaves the counter as it is
  hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
  hipError_t e = hipStreamIsCapturing(stream, &capture);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(g_scratch_mutex);  // wait, launch and record stay together: event order = array order
  StaticState& st = g_scratch[device];
  if (!st.ptr || !st.last_use) return hipErrorNotInitialized;
  if (capture != hipStreamCaptureStatusNone) {
    if ((e = hipMemsetAsync(st.ptr, 0, kScratchBytes, stream)) != hipSuccess) return e;
    return vkr_launch_cull_deterministic(p, st.ptr, kCullSeqRecorded, resident_blocks, fault, stream);
  }
  if ((e = hipStreamWaitEvent(stream, st.last_use, 0)) != hipSuccess) return e;  // (an event never recorded counts as done)
  const bool wrap = st.seq + 1u >= kCullSeqRecorded;
  if (wrap && (e = hipMemsetAsync(st.ptr, 0, kScratchBytes, stream)) != hipSuccess) return e;
  st.seq = wrap ? 1u : st.seq + 1u;
  e = vkr_launch_cull_deterministic(p, st.ptr, st.seq, resident_blocks, fault, stream);
  const hipError_t r = hipEventRecord(st.last_use, stream);  // also after a failed launch: whatever was enqueued is ordered
  return e != hipSuccess ? e : r;
}

hipError_t vkr_launch_cull_reset(const CullParams& p, hipStream_t stream) {
  if (p.mesh_count == 0u) return hipSuccess;
  hipLaunchKernelGGL(cull_init_cmds, dim3((p.mesh_count + 255u) / 256u), dim3(256), 0, stream, p);
  return hipGetLastError();
}

hipError_t vkr_launch_cull_atomic(const CullParams& p, bool commands_ready, hipStream_t stream) {
  if (p.mesh_count == 0u) return hipSuccess;
  const uint32_t chunks = (p.instance_count + kAtomicChunk - 1u) / kAtomicChunk;
  // 4 waves per block; cap the grid at 8 blocks per CU and let waves stride over chunks
  uint32_t blocks = (chunks + 3u) / 4u;
  if (blocks > 2048u) blocks = 2048u;
  if (!commands_ready) hipLaunchKernelGGL(cull_init_cmds, dim3((p.mesh_count + 255u) / 256u), dim3(256), 0, stream, p);
  if (p.instance_count == 0u) return hipGetLastError();
  hipLaunchKernelGGL(cull_atomic_kernel, dim3(blocks), dim3(kBlockThreads), 0, stream, p, chunks);
  return hipGetLastError();
}
