// k_select.hip -- K6: order-preserving compaction of the fixed-slot candidate table, score
// scatter and the score threshold.
//
// Replaces the list concatenation of HandSearch::evaluateHands (src/agile_grasp2/
// hand_search.cpp:223-228), the survivor list of GraspDetector::pruneGraspsOnHandParameters
// (grasp_detector.cpp:363-395; the predicate itself is evaluated inside k_sweep) and the
// score >= min_score_diff filter (grasp_detector.cpp:198-207).  Slot order (sample, orientation)
// IS the reference's output order, so a flag + exclusive scan + scatter keeps it.
#include <stddef.h>
#include <string.h>

#include "ag2_internal.h"

namespace ag2 {

// mode 0: slot holds a hypothesis; 1: hypothesis that survives the prune; 2: scored hypothesis
// with score >= thr
__global__ void k_slot_flags(const ag2_hypothesis* __restrict__ table,
                             const unsigned char* __restrict__ keep, int n_slots, int mode,
                             double thr, unsigned* __restrict__ flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n_slots) return;
  unsigned f = 0;
  if (i < n_slots) {
    // state byte: 0 empty, 1 survives the prune, 2 scored, 4 pruned away
    const bool valid = keep[i] != 0;  // (the records of empty slots are stale: never read them)
    if (mode == 0) f = valid;
    else if (mode == 1) f = valid && (keep[i] == 1 || keep[i] == 2);
    else f = valid && keep[i] == 2 && table[i].score >= thr;
  }
  flags[i] = f;
}

__global__ void k_slot_scatter(const unsigned* __restrict__ pref, int n_slots,
                               int* __restrict__ list) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_slots) return;
  if (pref[i + 1] != pref[i]) list[pref[i]] = i;
}

// The same in ONE launch for tables of up to 64 Ki slots (modes 0 and 1): each thread a contiguous run of
// 16 slots whose flags it keeps in a mask between the counting and the writing pass.  Five launches
// (flags, 3 x scan, scatter) and a copy become one; at S*R = 40 000 slots the table is latency-, not
// bandwidth-sized.
// Only the per-slot state byte is read (k_sweep: 0 empty, 1 hypothesis that survives the prune,
// 4 hypothesis pruned away), 16 slots per load, never the 176-byte records.
// With desc_off / desc_cnt the image descriptors (arena offset, point count) of the listed slots are
// written in the same pass (one launch less between the sweep and the renderer).
// With st_hyp the run's hypothesis statistics (count, points, largest list: what k_hyp_stats gathers
// from the same state bytes) are taken along: one launch less between the sweep and the read-back.
// Several 256-thread workgroups, each owning 4 096 consecutive slots (16 per thread: one 16-byte load).  A
// workgroup needs the number of listed slots in front of its chunk; with at most 16 chunks it simply counts
// them itself from the state bytes (<= 60 KB of L2-resident reads per workgroup) -- no inter-workgroup
// hand-over, no second launch.  (One 1 024-thread workgroup over all slots took 22 us at configuration 2:
// 40 slots per thread and a walk over up to ten occupied slots per thread, on one CU.)
constexpr int kCmpThreads = 256;
constexpr int kCmpChunk = kCmpThreads * 16;
__device__ __forceinline__ unsigned listed16(const uint4& v, int mode, int valid, unsigned* occ_out) {
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
  unsigned m = 0u, occ = 0u;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const unsigned b = (w[k >> 2] >> (8 * (k & 3))) & 255u;
    const bool in = k < valid;
    const bool f = in && (mode == 1 ? (b == 1u) : (b != 0u));
    m |= f ? (1u << k) : 0u;
    occ |= (in && b != 0u) ? (1u << k) : 0u;
  }
  if (occ_out) *occ_out = occ;
  return m;
}
__global__ void __launch_bounds__(kCmpThreads) k_compact_small(const unsigned char* __restrict__ keep,
                                                               int n_slots, int mode, int* __restrict__ list,
                                                               unsigned* __restrict__ count,
                                                               const ag2_hypothesis* __restrict__ table,
                                                               const long long* __restrict__ tab_off,
                                                               long long* __restrict__ desc_off,
                                                               int* __restrict__ desc_cnt, DevStats* st_hyp) {
  __shared__ unsigned wsum[4], wcnt[4], wmax[4], wbef[4];
  __shared__ unsigned long long wpts[4];
  const int t = threadIdx.x;
  const int w0 = blockIdx.x * kCmpChunk;  // (all slots in front of it exist: w0 <= n_slots)
  // listed slots in front of this workgroup's chunk
  unsigned before = 0u;
  for (int i = t * 16; i < w0; i += kCmpChunk)
    before += (unsigned)__popc(listed16(*reinterpret_cast<const uint4*>(keep + i), mode, 16, nullptr));
  before = (unsigned)wave_sum_i((int)before);
  if (lane_id() == 0) wbef[wave_id()] = before;
  // this thread's 16 slots (the buffer is 16-byte padded past n_slots: DevBuf slack)
  const int s0 = w0 + t * 16;
  unsigned mask = 0u, occ = 0u;
  if (s0 < n_slots) mask = listed16(*reinterpret_cast<const uint4*>(keep + s0), mode, min(16, n_slots - s0), &occ);
  const unsigned tot = (unsigned)__popc(mask);
  unsigned inc = tot;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned v = (unsigned)__shfl_up((int)inc, o, 64);
    if (lane_id() >= o) inc += v;
  }
  if (lane_id() == 63) wsum[wave_id()] = inc;
  __syncthreads();
  unsigned woff = 0, all = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    if (w < wave_id()) woff += wsum[w];
    all += wsum[w];
  }
  const unsigned base = wbef[0] + wbef[1] + wbef[2] + wbef[3];
  unsigned pos = base + woff + inc - tot;
  // ONE walk over the occupied slots of the thread's run, four at a time with their gathers in flight
  // together (few slots are occupied; one slot per step was a chain of dependent round trips): the
  // point count feeds the statistics, and for a listed slot the descriptor as well.
  unsigned hc = 0, hm = 0;
  unsigned long long hp = 0;
  const unsigned walk = st_hyp ? occ : mask;
  for (unsigned m = walk; m;) {
    int sl[4];
    unsigned p[4];
    long long off[4];
    bool listed[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int b = m ? __ffs((int)m) - 1 : 0;
      sl[u] = m ? s0 + b : -1;
      listed[u] = m && ((mask >> b) & 1u);
      p[u] = (sl[u] >= 0 && (st_hyp || desc_off)) ? (unsigned)table[sl[u]].n_points : 0u;
      off[u] = (listed[u] && desc_off) ? tab_off[sl[u]] : 0ll;
      m &= m - 1u;  // (0 stays 0)
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (sl[u] >= 0) {
        hc++;
        hp += p[u];
        hm = max(hm, p[u]);
      }
      if (listed[u]) {
        list[pos] = sl[u];
        if (desc_off) {
          desc_off[pos] = off[u];
          desc_cnt[pos] = (int)p[u];
        }
        pos++;
      }
    }
  }
  if (st_hyp) {  // uniform
    hc = (unsigned)wave_sum_i((int)hc);
    const unsigned plo = (unsigned)wave_sum_i((int)(unsigned)(hp & 0xFFFFFFu));  // < 2^24 per lane: no overflow
    const unsigned phi = (unsigned)wave_sum_i((int)(unsigned)(hp >> 24));
    hm = (unsigned)(-wave_min_i(-(int)hm));
    if (lane_id() == 0) {
      wcnt[wave_id()] = hc;
      wpts[wave_id()] = (unsigned long long)plo + ((unsigned long long)phi << 24);
      wmax[wave_id()] = hm;
    }
    __syncthreads();
  }
  if (t == 0) {
    if (blockIdx.x == gridDim.x - 1) *count = base + all;  // the last chunk knows the total
    if (st_hyp) {
      unsigned c = 0, mx = 0;
      unsigned long long pp = 0;
      for (int w = 0; w < 4; w++) {
        c += wcnt[w];
        pp += wpts[w];
        mx = max(mx, wmax[w]);
      }
      if (c) {  // (added, like k_hyp_stats)
        atomicAdd(&st_hyp->n_hyp, c);
        atomicAdd(&st_hyp->sum_p, pp);
        atomicMax(&st_hyp->max_p, mx);
      }
    }
  }
}

// Asynchronous: the list length is left on the device (*d_count) -- no host round trip.
int compact_slots_async(ag2_ctx* c, size_t n_slots, int mode, DevBuf& out_list, unsigned* d_count,
                        bool with_descs) {
  c->desc_stride = 0;
  // statistics the split sweep left to be gathered from the slot table: along with the small
  // compaction when that runs, by their own kernel otherwise
  const bool stats_due = c->hyp_stats_pending;
  const bool small = n_slots <= 65536 && mode <= 1;
  c->hyp_stats_pending = false;
  if (n_slots == 0) {
    AG2_HIP(c, hipMemsetAsync(d_count, 0, 4, c->stream));
    return 0;
  }
  if (stats_due && !small) {
    const int rc = launch_hyp_stats(c, n_slots);
    if (rc) return rc;
  }
  AG2_HIP(c, out_list.reserve(n_slots * 4));
  if (small) {
    long long* d_off = nullptr;
    if (with_descs) {  // descriptors for up to n_slots images: offsets, then counts
      AG2_HIP(c, c->d_desc.reserve(n_slots * 12));
      d_off = c->d_desc.as<long long>();
      c->desc_stride = n_slots;
    }
    hipLaunchKernelGGL(k_compact_small, dim3((unsigned)((n_slots + kCmpChunk - 1) / kCmpChunk)), dim3(kCmpThreads), 0, c->stream,
                       c->d_tab_keep.as<unsigned char>(), (int)n_slots, mode, out_list.as<int>(), d_count,
                       c->d_table.as<ag2_hypothesis>(), c->d_tab_off.as<long long>(), d_off,
                       d_off ? (int*)(d_off + n_slots) : (int*)nullptr,
                       stats_due ? c->d_stats.as<DevStats>() : (DevStats*)nullptr);
    AG2_HIP(c, hipGetLastError());
    return 0;
  }
  AG2_HIP(c, c->d_flags.reserve((n_slots + 1) * 4));
  unsigned* fl = c->d_flags.as<unsigned>();
  const int nb = ((int)n_slots + 1 + 255) / 256;
  hipLaunchKernelGGL(k_slot_flags, dim3(nb), dim3(256), 0, c->stream,
                     c->d_table.as<ag2_hypothesis>(), c->d_tab_keep.as<unsigned char>(),
                     (int)n_slots, mode, c->p.min_score_diff, fl);
  const int rc = scan_exclusive_u32(c, fl, (int)n_slots + 1);
  if (rc) return rc;
  hipLaunchKernelGGL(k_slot_scatter, dim3(nb), dim3(256), 0, c->stream, fl, (int)n_slots,
                     out_list.as<int>());
  AG2_HIP(c, hipMemcpyAsync(d_count, fl + n_slots, 4, hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

int compact_slots(ag2_ctx* c, size_t n_slots, int mode, DevBuf& out_list, size_t* n_out) {
  *n_out = 0;
  if (n_slots == 0) return 0;
  unsigned* d_count = &c->d_stats.as<DevStats>()->n_list;
  const int rc = compact_slots_async(c, n_slots, mode, out_list, d_count, false);
  if (rc) return rc;
  unsigned total = 0;
  AG2_HIP(c, hipMemcpyAsync(&total, d_count, 4, hipMemcpyDeviceToHost, c->stream));
  AG2_HIP(c, hipStreamSynchronize(c->stream));
  *n_out = total;
  return 0;
}

// ---- tail of detectGraspPoses: score, threshold, ordered gather (grasp_detector.cpp:198-207) ----
// per scored image i (list order = output order): score = ip2[1] - ip2[0] (:200) written into the
// table slot, flag = score >= min_score_diff (:202)
// d_n (frame mode): n is the list's capacity, *d_n its length; flags beyond the length are zero so
// that the scan over the whole capacity leaves the total in pref[capacity].
__global__ void k_score_flags(const float* __restrict__ logits, const int* __restrict__ list, int n,
                              const unsigned* __restrict__ d_n,
                              ag2_hypothesis* __restrict__ table, unsigned char* __restrict__ keep,
                              double thr, unsigned* __restrict__ flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  if (d_n && i >= (int)*d_n) {
    flags[i] = 0;
    return;
  }
  unsigned f = 0;
  if (i < n) {
    const int s = list[i];
    const float sc = logits[2 * i + 1] - logits[2 * i];
    table[s].score = (double)sc;
    keep[s] = 2;
    f = ((double)sc >= thr) ? 1u : 0u;
  }
  flags[i] = f;
}

__global__ void k_gather_selected(const unsigned* __restrict__ pref, const int* __restrict__ list,
                                  int n, const ag2_hypothesis* __restrict__ table,
                                  ag2_hypothesis* __restrict__ out, unsigned* __restrict__ count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    *count = pref[n];
    *reinterpret_cast<unsigned*>(out + n) = pref[n];  // trailer: one copy brings records + count
  }
  if (i >= n) return;
  if (pref[i + 1] != pref[i]) {
    ag2_hypothesis h = table[list[i]];
    h.full_antipodal = 1;  // grasp_detector.cpp:205
    out[pref[i]] = h;
  }
}

// The same three steps for the usual few hundred to few thousand scored images in ONE launch of one
// workgroup (five launches of ~4 us each otherwise): thread t owns kSelPer consecutive list entries,
// so a block scan of the per-thread counts gives every selected record its position in list order.
constexpr int kSelThreads = 1024, kSelPer = 8, kSelSmall = kSelThreads * kSelPer;
__global__ void __launch_bounds__(kSelThreads) k_select_small(
    const float* __restrict__ logits, const int* __restrict__ list, int n, const unsigned* __restrict__ d_n,
    ag2_hypothesis* __restrict__ table,
    unsigned char* __restrict__ keep, double thr, ag2_hypothesis* __restrict__ out,
    unsigned* __restrict__ count) {
  __shared__ unsigned wsum[kSelThreads / kWave];
  __shared__ int sel_slot[kSelSmall];
  const int n_cap = n;  // the count trailer sits behind the list's capacity
  if (d_n) n = min(n, (int)*d_n);
  const int tid = threadIdx.x;
  int slot[kSelPer];
  unsigned sel = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < kSelPer; k++) {
    const int i = tid * kSelPer + k;
    slot[k] = 0;
    if (i < n) {
      const int s = list[i];
      const float sc = logits[2 * i + 1] - logits[2 * i];
      table[s].score = (double)sc;
      keep[s] = 2;
      slot[k] = s;
      if ((double)sc >= thr) {
        sel |= 1u << k;
        tot++;
      }
    }
  }
  unsigned inc = tot;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = (unsigned)__shfl_up((int)inc, o, 64);
    if (lane_id() >= o) inc += t;
  }
  if (lane_id() == 63) wsum[wave_id()] = inc;
  __syncthreads();
  unsigned woff = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kSelThreads / kWave; w++) {
    if (w < wave_id()) woff += wsum[w];
    total += wsum[w];
  }
  unsigned run = woff + inc - tot;
#pragma unroll
  for (int k = 0; k < kSelPer; k++)
    if (sel & (1u << k)) sel_slot[run++] = slot[k];
  __syncthreads();  // the slots of the selected records in output order; every score is in the table
  // the records are copied by all threads, 16 bytes each (a thread copying its own up to eight
  // 176-byte records one after the other was most of this kernel's time)
  constexpr int kParts = (int)(sizeof(ag2_hypothesis) / 16);
  static_assert(sizeof(ag2_hypothesis) == 176 && offsetof(ag2_hypothesis, full_antipodal) == 169, "record layout");
  const uint4* src = reinterpret_cast<const uint4*>(table);
  uint4* dst = reinterpret_cast<uint4*>(out);
  for (int e = tid; e < (int)total * kParts; e += kSelThreads) {
    const int q = e / kParts, part = e - q * kParts;
    uint4 v = src[(size_t)sel_slot[q] * kParts + part];
    if (part == kParts - 1) v.z = (v.z & ~0xFF00u) | 0x0100u;  // full_antipodal = 1 (grasp_detector.cpp:205)
    dst[(size_t)q * kParts + part] = v;
  }
  if (tid == 0) {
    *count = total;
    *reinterpret_cast<unsigned*>(out + n_cap) = total;  // trailer: one copy brings records + count
  }
}

// Leaves the selected records (score >= threshold, list order) in d_sel and their count in *d_count.
// d_n (frame mode): n_img is the capacity of the list, its length is read from *d_n on the device.
int score_and_select_async(ag2_ctx* c, const int* d_list, size_t n_img, unsigned* d_count,
                           const unsigned* d_n) {
  if (n_img == 0) {
    AG2_HIP(c, hipMemsetAsync(d_count, 0, 4, c->stream));
    return 0;
  }
  AG2_HIP(c, c->d_sel.reserve(n_img * sizeof(ag2_hypothesis) + 16));  // + the count trailer
  if (n_img <= (size_t)kSelSmall) {
    hipLaunchKernelGGL(k_select_small, dim3(1), dim3(kSelThreads), 0, c->stream, c->d_logits.as<float>(),
                       d_list, (int)n_img, d_n, c->d_table.as<ag2_hypothesis>(),
                       c->d_tab_keep.as<unsigned char>(), c->p.min_score_diff,
                       c->d_sel.as<ag2_hypothesis>(), d_count);
    AG2_HIP(c, hipGetLastError());
    return 0;
  }
  AG2_HIP(c, c->d_flags.reserve((n_img + 1) * 4));
  unsigned* fl = c->d_flags.as<unsigned>();
  const int nb = ((int)n_img + 1 + 255) / 256;
  hipLaunchKernelGGL(k_score_flags, dim3(nb), dim3(256), 0, c->stream, c->d_logits.as<float>(), d_list,
                     (int)n_img, d_n, c->d_table.as<ag2_hypothesis>(), c->d_tab_keep.as<unsigned char>(),
                     c->p.min_score_diff, fl);
  const int rc = scan_exclusive_u32(c, fl, (int)n_img + 1);
  if (rc) return rc;
  hipLaunchKernelGGL(k_gather_selected, dim3(nb), dim3(256), 0, c->stream, fl, d_list, (int)n_img,
                     c->d_table.as<ag2_hypothesis>(), c->d_sel.as<ag2_hypothesis>(), d_count);
  AG2_HIP(c, hipGetLastError());
  return 0;
}

// image descriptors (arena offset, point count) of the listed slots
__global__ void k_image_descs(const ag2_hypothesis* __restrict__ table,
                              const long long* __restrict__ tab_off, const int* __restrict__ list,
                              int n, long long* __restrict__ desc_off, int* __restrict__ desc_cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = list[i];
  desc_off[i] = tab_off[s];
  desc_cnt[i] = table[s].n_points;
}


__global__ void k_gather_records(const ag2_hypothesis* __restrict__ table,
                                 const long long* __restrict__ tab_off,
                                 const unsigned char* __restrict__ keep,
                                 const int* __restrict__ list, int n,
                                 ag2_hypothesis* __restrict__ out, long long* __restrict__ out_off,
                                 unsigned char* __restrict__ out_keep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = list[i];
  out[i] = table[s];
  if (out_off) out_off[i] = tab_off[s];
  if (out_keep) out_keep[i] = (keep[s] == 1 || keep[s] == 2) ? 1 : 0;  // 4 = pruned away
}

int gather_records(ag2_ctx* c, const int* d_list, size_t n, std::vector<ag2_hypothesis>& recs,
                   std::vector<int64_t>* offs, std::vector<uint8_t>* keep) {
  recs.resize(n);
  if (offs) offs->resize(n);
  if (keep) keep->resize(n);
  if (n == 0) return 0;
  const size_t bytes = n * (sizeof(ag2_hypothesis) + 8 + 1) + 64;
  AG2_HIP(c, c->d_tmp.reserve(bytes));
  ag2_hypothesis* d_rec = c->d_tmp.as<ag2_hypothesis>();
  long long* d_off = (long long*)(d_rec + n);
  unsigned char* d_keep = (unsigned char*)(d_off + n);
  hipLaunchKernelGGL(k_gather_records, dim3(((int)n + 255) / 256), dim3(256), 0, c->stream,
                     c->d_table.as<ag2_hypothesis>(), c->d_tab_off.as<long long>(),
                     c->d_tab_keep.as<unsigned char>(), d_list, (int)n, d_rec, d_off, d_keep);
  AG2_HIP(c, hipMemcpyAsync(recs.data(), d_rec, n * sizeof(ag2_hypothesis), hipMemcpyDeviceToHost, c->stream));
  if (offs) AG2_HIP(c, hipMemcpyAsync(offs->data(), d_off, n * 8, hipMemcpyDeviceToHost, c->stream));
  if (keep) AG2_HIP(c, hipMemcpyAsync(keep->data(), d_keep, n, hipMemcpyDeviceToHost, c->stream));
  AG2_HIP(c, hipStreamSynchronize(c->stream));
  return 0;
}

// ---- compact form of the candidate table for the exchange step -------------------------------
// header (16 B): {count, cap, 0, 0}; then min(count, cap) records in slot order.  One thread per 16
// bytes; the count is read on the device (written by the compaction just before).
__global__ void k_export_compact(const uint4* __restrict__ table, const int* __restrict__ list,
                                 unsigned cap, uint4* __restrict__ dst) {
  constexpr unsigned kPer = (unsigned)(sizeof(ag2_hypothesis) / 16);
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned count = reinterpret_cast<const unsigned*>(dst)[0];
  const unsigned n = count < cap ? count : cap;
  if (i == 0) reinterpret_cast<unsigned*>(dst)[1] = cap;  // (words 2, 3 cleared by the caller)
  const unsigned r = i / kPer, k = i % kPer;
  if (r < n) dst[1 + i] = table[(size_t)list[r] * kPer + k];
}

int export_candidates_compact(ag2_ctx* c, void* d_dst, size_t cap_records) {
  const size_t n_slots = c->s * (size_t)c->p.num_orientations;
  AG2_HIP(c, hipMemsetAsync(d_dst, 0, 16, c->stream));
  if (n_slots == 0 || cap_records == 0) return 0;
  // every hypothesis (slot state != 0), slot order; the count lands in the header's first word
  const int rc = compact_slots_async(c, n_slots, 0, c->d_export_list, (unsigned*)d_dst);
  if (rc) return rc;
  const size_t threads = cap_records * (sizeof(ag2_hypothesis) / 16);
  hipLaunchKernelGGL(k_export_compact, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, c->stream,
                     c->d_table.as<uint4>(), c->d_export_list.as<int>(), (unsigned)cap_records, (uint4*)d_dst);
  AG2_HIP(c, hipGetLastError());
  return 0;
}

// ---- multi-GPU merge: every rank's selected list travels (compact form), every rank ranks them all ----
// header (16 B): {count, cap, 0, 0}; then min(count, cap) records in list order
// Header of a rank's list: {count, cap, status, images scored}.  status bit 0: the rank ran its detect at shapes
// learned from its previous call (ag2_ctx::RankSpec) and they did not hold -- more images than the tail was
// launched for, a longer in-box list than its renderers take, a sweep buffer too small, a sample for the
// long-list stage that was left out: its list is then void (count 0) and every rank repeats the step.
struct ExportSpec {
  const DevStats* st;        // NULL: no statistics to report (header {count, cap, 0, 0})
  DevStats* st_host;         // page-locked copy of the statistics for the rank's host (or NULL)
  unsigned check;            // 1: the shapes below are to be checked against *st
  unsigned cap_img;
  int render_cap;
  int stage1_skipped;
};
__global__ void k_export_selected(const uint4* __restrict__ recs, const unsigned* __restrict__ d_count,
                                  unsigned cap, uint4* __restrict__ dst, ExportSpec es) {
  constexpr unsigned kPer = (unsigned)(sizeof(ag2_hypothesis) / 16);
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned status = 0u, n_scored = 0u;
  if (es.st) {  // (uniform)
    n_scored = es.st->n_list;
    if (es.check)
      status = ((es.st->err_flags & (1u | 2u | 8u)) != 0u || n_scored > es.cap_img || (int)es.st->max_p > es.render_cap ||
                (es.stage1_skipped && es.st->n_overflow > 0u)) ? 1u : 0u;
  }
  const unsigned count = status ? 0u : *d_count;
  const unsigned n = count < cap ? count : cap;
  if (i == 0) {
    dst[0] = make_uint4(count, cap, status, n_scored);
    if (es.st && es.st_host) *es.st_host = *es.st;
  }
  if (i / kPer < n) dst[1 + i] = recs[i];
}

int export_selected_compact(ag2_ctx* c, void* d_dst, size_t cap_records) {
  const size_t threads = std::max<size_t>(cap_records, 1) * (sizeof(ag2_hypothesis) / 16);
  ExportSpec es{};
  if (c->rank_spec.pending && c->h_pin_dev) {
    es.st = c->d_stats.as<DevStats>();
    es.st_host = reinterpret_cast<DevStats*>(pin_small_dev(c) + kPinRankStats);
    es.check = 1u;
    es.cap_img = c->rank_spec.cap_img;
    es.render_cap = c->rank_spec.render_cap;
    es.stage1_skipped = c->rank_spec.stage1_skipped;
  }
  hipLaunchKernelGGL(k_export_selected, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, c->stream,
                     (const uint4*)c->d_last_sel, c->d_last_nsel, (unsigned)cap_records, (uint4*)d_dst, es);
  AG2_HIP(c, hipGetLastError());
  return 0;
}

// The gathered buffers (world x (16 B header + cap records)) flattened in (rank, list position) order --
// rank order is sample order, so this is the order the unsplit run's list has -- then the top
// num_selected by score, ties by position (grasp_detector.cpp:239-252), by rank counting as in the frame
// path.  One workgroup per 256 records; out[0 .. k) in rank order, the counts behind them.
__global__ void __launch_bounds__(256) k_merge_flatten(const unsigned char* __restrict__ g, int world, unsigned cap,
                                                      ag2_hypothesis* __restrict__ flat, unsigned* __restrict__ total) {
  const size_t per = 16 + (size_t)cap * sizeof(ag2_hypothesis);
  const int r = blockIdx.y;
  unsigned off = 0, mine = 0, cut = 0;
  for (int k = 0; k <= r; k++) {  // (a few ranks: every thread adds up the counts before its rank)
    const unsigned* hp = reinterpret_cast<const unsigned*>(g + (size_t)k * per);
    const unsigned hdr = hp[0];
    const unsigned cnt = min(hdr, cap);
    cut |= (hdr > cap) ? 1u : 0u;  // a rank's list was cut at the exchange capacity: the merge is void
    cut |= (hp[2] & 1u) << 1;      // a rank's one-trip detect did not hold its shapes: every rank repeats the step
    if (k < r) off += cnt; else mine = cnt;
  }
  if (r == world - 1 && blockIdx.x == 0 && threadIdx.x == 0) {
    total[0] = off + mine;
    total[1] = cut;
  }
  constexpr unsigned kPer = (unsigned)(sizeof(ag2_hypothesis) / 16);
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i / kPer < mine)
    reinterpret_cast<uint4*>(flat + off)[i] = reinterpret_cast<const uint4*>(g + (size_t)r * per + 16)[i];
}

__global__ void __launch_bounds__(256) k_merge_topk(const ag2_hypothesis* __restrict__ recs,
                                                   const unsigned* __restrict__ d_n, int k_want, int k_cap,
                                                   ag2_hypothesis* __restrict__ out, unsigned* __restrict__ out_counts) {
  __shared__ double sc[256];
  const int n = (int)*d_n;
  int k = (k_want >= 0 && k_want < n) ? k_want : n;
  k = min(k, k_cap);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out_counts[0] = (unsigned)k;
    out_counts[1] = (unsigned)n;
  }
  const int base = blockIdx.x * 256;
  if (base >= n) return;  // uniform
  const int i = base + threadIdx.x;
  __shared__ unsigned long long keys[kRankKeys];
  __shared__ int inexact;
  int rank = rank_by_keys(recs, n, i, keys, &inexact);
  const bool general = rank < 0;  // (uniform) longer than the key stage, or a score that is no float
  const double si = (general && i < n) ? recs[i].score : 0.0;
  if (general) rank = 0;
  for (int j0 = 0; general && j0 < n; j0 += 256) {
    __syncthreads();
    sc[threadIdx.x] = (j0 + (int)threadIdx.x < n) ? recs[j0 + threadIdx.x].score : -__builtin_inf();
    __syncthreads();
    // (scores beyond n are padded so that they never count.)  Eight LDS reads in flight and no branch:
    // one read per iteration, each waited for, behind a short-circuit compare made this loop the
    // whole kernel.
    for (int t = 0; t < 256; t += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = sc[t + u];
#pragma unroll
      for (int u = 0; u < 8; u++)
        rank += (int)(v[u] > si) | ((int)(v[u] == si) & (int)(j0 + t + u < i));
    }
  }
  if (i < n && rank < k) out[rank] = recs[i];
}

int merge_selected(ag2_ctx* c, const void* d_gathered, size_t world, size_t cap_records, ag2_hypothesis* selected,
                   size_t cap, size_t* n_selected, size_t* n_total) {
  const size_t n_max = world * cap_records;
  const size_t k_cap = (c->p.num_selected >= 0) ? std::min<size_t>((size_t)c->p.num_selected, n_max) : n_max;
  AG2_HIP(c, c->d_merge.reserve((n_max + k_cap + 1) * sizeof(ag2_hypothesis) + 64));
  ag2_hypothesis* flat = c->d_merge.as<ag2_hypothesis>();
  ag2_hypothesis* out = flat + n_max;
  unsigned* counts = reinterpret_cast<unsigned*>(out + k_cap);  // {k, n}, then {flat total, a list was cut}, then the clustered count
  const size_t threads = std::max<size_t>(cap_records, 1) * (sizeof(ag2_hypothesis) / 16);
  hipLaunchKernelGGL(k_merge_flatten, dim3((unsigned)((threads + 255) / 256), (unsigned)world), dim3(256), 0, c->stream,
                     (const unsigned char*)d_gathered, (int)world, (unsigned)cap_records, flat, counts + 2);
  // Grasp clusters (grasp_detector.cpp:228-236) count inliers over ALL hands that passed the threshold,
  // whichever rank found them: the ranks export their lists BEFORE the clustering and it runs here, on
  // the gathered list (rank order = sample order = the order the unsplit run clusters in).
  const ag2_hypothesis* recs = flat;
  const unsigned* d_nrecs = counts + 2;
  if (c->min_inliers > 0 && n_max > 0) {
    const int rc = cluster_async(c, flat, n_max, counts + 2, c->min_inliers, counts + 4);
    if (rc) return rc;
    recs = c->d_cluster.as<ag2_hypothesis>();
    d_nrecs = counts + 4;
  }
  hipLaunchKernelGGL(k_merge_topk, dim3((unsigned)((n_max + 255) / 256)), dim3(256), 0, c->stream, recs, d_nrecs,
                     c->p.num_selected, (int)k_cap, out, counts);
  AG2_HIP(c, hipGetLastError());
  // one copy brings the top-k records and the counts behind them
  const size_t bytes = k_cap * sizeof(ag2_hypothesis) + 16;
  int rc = pin_reserve(c, bytes);
  if (rc) return rc;
  AG2_HIP(c, hipMemcpyAsync(pin_bulk(c), out, bytes, hipMemcpyDeviceToHost, c->stream));
  AG2_HIP(c, hipStreamSynchronize(c->stream));
  unsigned kn[4];
  __builtin_memcpy(kn, pin_bulk(c) + k_cap * sizeof(ag2_hypothesis), 16);
  *n_selected = kn[0];
  if (n_total) *n_total = kn[2];  // records that took part (before the clustering)
  {  // (the stream is idle: this rank's own one-trip detect, if any, has left its statistics)
    const int rcc = rank_spec_collect(c, /*stream_is_idle=*/true);
    if (rcc) return rcc;
  }
  if (kn[3] & 2u) {
    *n_selected = 0;
    c->spec_cap_img = 0;  // this rank's next ag2_detect runs step by step (and learns its shapes again)
    return set_err(c, AG2_ERR_RETRY, "merge: a rank's detect ran at shapes that did not hold (its header says so): "
                                     "nothing was merged; every rank repeats the step");
  }
  if (kn[3] & 1u)
    return set_err(c, AG2_ERR_CAPACITY, "merge: a rank's list is longer than the exchange capacity (header count > "
                                        "cap_records): the global top-k would be wrong; exchange with a larger capacity");
  if (kn[0] > cap) return set_err(c, AG2_ERR_CAPACITY, "merge: output capacity too small");
  if (kn[0] && selected) __builtin_memcpy(selected, pin_bulk(c), (size_t)kn[0] * sizeof(ag2_hypothesis));
  return 0;
}

int make_image_descs(ag2_ctx* c, const int* d_list, size_t n) {
  AG2_HIP(c, c->d_desc.reserve(std::max<size_t>(n, 1) * 12));
  if (n == 0) return 0;
  long long* d_off = c->d_desc.as<long long>();
  int* d_cnt = (int*)(d_off + n);
  hipLaunchKernelGGL(k_image_descs, dim3(((int)n + 255) / 256), dim3(256), 0, c->stream,
                     c->d_table.as<ag2_hypothesis>(), c->d_tab_off.as<long long>(), d_list, (int)n,
                     d_off, d_cnt);
  AG2_HIP(c, hipGetLastError());
  return 0;
}


}  // namespace ag2
