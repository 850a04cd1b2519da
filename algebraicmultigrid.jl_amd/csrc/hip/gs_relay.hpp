// gs_relay.hpp — the dataflow sweep of gs_flow.hpp with a block's walk RELAYED between several waves.
// Included by gs_schedule.hpp (single-column sweeps of block-ordered levels) and tools/block_wave_bench.hip.
//
// smoother.jl:61-90 (gs!) / :193-221 (sor_step!) in exact lexicographic order: the same plan, records, mailboxes, fetcher
// and arithmetic as gs_bw_flow_kernel — what changes is WHO walks.  A single wave64 pays ~3 ns per instruction whatever it
// is (tools/wave_chain_probe), and a step of the walk is ~110 (7-point rows) to ~250 (19-point rows) instructions of which
// only the tail — gather the x of the rows just computed, the ordered sum, the quotient, the LDS store — depends on the
// previous step: the walk of gs_bw_flow_kernel is bound by what one wave can ISSUE (0.24 / 0.38 us per step with the
// operands in LDS, 0.40 / 0.50 us streamed), not by the chain of dependent operations (~0.07 us).  Here W walker waves take
// the steps of a block in turn (step k belongs to wave k mod W): a wave unpacks its step's record, computes its LDS
// addresses and its mailbox address, issues the loads of its step after next, all of it while the W - 1 steps in front of
// its own are being computed by the other waves (each on its own SIMD's issue port); then it waits for the block's DONE word
// in LDS (steps finished so far, written behind the x of a step by the wave that computed it — LDS operations of one
// wave execute in program order, so whoever reads the word sees the values), runs the dependent tail, stores x and the word.
// Per step the critical path is the hand-over (one LDS round trip) + the tail.  W x D steps of records are in flight in
// registers (D sets per wave), which also covers the HBM latency the single walker's D = 3-4 steps did not.
// Everything else — tickets in depth order, publish-as-computed mailboxes, the fetcher wave, bounded polls that raise
// *err — is gs_flow.hpp's.  Per row the arithmetic is unchanged: bitwise the scalar loop.
// Template flag DICT: the records on the dictionary layout (gs_flow.hpp FlowDict) — a set in flight is the row's column chunks
// and b, the values come out of the block's dictionary in LDS ahead of the hand-over (256^3: 0.61 -> 0.56 / 0.93 -> 0.78 ms).
#pragma once
#include "gs_flow.hpp"

namespace amgh {
namespace bw {

#ifndef BW_RELAY_W
#define BW_RELAY_W 3                    // walker waves per block (256^3, levels 0 / 1, ms per sweep: 2: 0.625 / 0.96, 3: 0.610 / 0.93, 4: 0.73 / 1.00; one walker: 0.66 / 0.98)
#endif

// walker waves / register sets per wave / waves per SIMD the kernel is compiled for
#ifndef BW_RELAY_SPIN
#define BW_RELAY_SPIN 48u               // polls of the hand-over before a walker sleeps between them
#endif
#ifndef BW_RELAY_DEPTH_SHORT
#define BW_RELAY_DEPTH_SHORT 3          // register sets per walker wave, rows of <= 12 entries ...
#endif
#ifndef BW_RELAY_DEPTH_LONG
#define BW_RELAY_DEPTH_LONG 2           // ... and of up to 18
#endif
template <int MAXK> struct RelayDepth { static constexpr int value = MAXK <= 12 ? BW_RELAY_DEPTH_SHORT : BW_RELAY_DEPTH_LONG; };
#ifndef BW_RELAY_FETCH_SLEEP
#define BW_RELAY_FETCH_SLEEP 4          // x 64 cycles between two polls of a fetcher that saw nothing new
#endif
#ifndef BW_RELAY_FETCH_PD
#define BW_RELAY_FETCH_PD 1             // polls of a fetcher in flight at once (1: the round-5 loop — wait for a poll, then issue the next)
#endif
#ifndef BW_RELAY_FETCH_GAP
#define BW_RELAY_FETCH_GAP 6            // ... issued this many x 64 cycles apart
#endif
#ifndef BW_RELAY_DICT_DEPTH
#define BW_RELAY_DICT_DEPTH 3
#endif
template <int MAXK> struct RelayDepthD { static constexpr int value = BW_RELAY_DICT_DEPTH; };   // the dictionary layout: a set is a handful of registers
template <int MAXK> struct RelayWaves { static constexpr int value = MAXK <= 6 ? 4 : MAXK <= 18 ? 2 : 1; };
// (the dictionary layout: a set in flight is a column chunk or three and b — 74 / 139 registers at 6 / 18 entries a row against 109 / 165)
template <int MAXK> struct RelayWavesD { static constexpr int value = MAXK <= 6 ? 5 : MAXK <= 18 ? 3 : 2; };

// LDS by its 32-bit address (the walkers keep absolute LDS addresses in registers: nothing is added behind the hand-over)
template <typename T> __device__ __forceinline__ T lds_get(unsigned addr) { return *(const __attribute__((address_space(3))) T*)(unsigned long long)addr; }
template <typename T> __device__ __forceinline__ void lds_put(unsigned addr, T v) { *(__attribute__((address_space(3))) T*)(unsigned long long)addr = v; }
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) unsigned char*)p; }

// control words of a block, behind its x in LDS (16 bytes, 16-byte aligned): one ds_read_b128 polls them all
//   [0] progress: leading entries of the fetch list that have arrived (the fetcher)   [1] done: steps finished (the walkers)
//   [2] bail: somebody gave up a poll — nobody waits any more                          [3] unused
// One block of the sweep: ticket ut of the launch, mailbox tag `epoch`.  All waves of the workgroup enter; the fetcher wave
// leaves when its list is done, the walkers behind their last step.
// Template flag LATE (round 6): the dependency-aware row sum.  The records hold a row's entries split around the padding — the
// columns that precede the row in slots [0, MAXK / 2), the columns behind it in [MAXK / 2, MAXK) (gs_blocks.hpp, Plan::late_ok) —
// so the half of the sum that multiplies the sweep's FAR side (x values nobody writes before this row is done: rows behind it
// in a forward sweep, before it in a backward one) is gathered and summed ABOVE the hand-over, and only the near half — the
// rows the previous steps and the predecessor blocks produce — is gathered and added below it: MAXK / 2 gathers, products and
// dependent adds in the tail instead of MAXK (a wave pays ~3 ns per instruction, whatever it is).  Per row: early = sum of
// the far half in stored order, acc = early + the near half in stored order — deterministic, the same Gauss-Seidel iterate,
// not the scalar loop's bits (tunable gs_bw_inorder = 1 keeps those).
template <typename R, bool SOR, bool BWD, int MAXK, int W, bool DICT, bool LATE>
__device__ __forceinline__ void relay_block(const FlowArgs<R>& a, unsigned char* lds_all, const int wv, const int lane, const unsigned int ut,
                                            const unsigned int epoch, const long long t_start) {
  typedef typename std::conditional<DICT, FlowOpsD<R, MAXK>, FlowOps<R, MAXK>>::type O;
  typedef Mail<R> M;
  constexpr int D = DICT ? RelayDepthD<MAXK>::value : RelayDepth<MAXK>::value;
  constexpr int RB = (int)sizeof(R);
  constexpr int NVC = FlowOps<R, MAXK>::NVC;
  const bool fetcher = wv == W;
  const int tk = (int)ut;
  const int ob = __builtin_amdgcn_readfirstlane(BWD ? a.nblocks - 1 - tk : tk);
  const Desc d = a.blocks[ob];
  const FlowDesc f = a.fd[ob];
  const int ns = d.nlev;
  const int nxb = (int)(((size_t)(d.nrows + d.next + 1) * sizeof(R) + 15) & ~(size_t)15);
  const unsigned spin_limit = a.spin_limit ? a.spin_limit : (1u << 22);
  unsigned char* lds = lds_all;
  R* xl = (R*)lds;
  unsigned int* ctl = (unsigned int*)(lds + nxb);
  R* x = a.x;
  if (fetcher) {
    // ---- load phase, the fetcher's half: the external columns of the far side as they stand (gs_flow.hpp) ----
    constexpr int EU = 4;
    const R* xc = uniform_ptr(x);
    const int nfar = BWD ? d.npre + (d.next - d.npre - f.npost) : d.next - d.npre;
    if (a.xzero) {
      for (int e0 = 0; e0 < d.next; e0 += 64) { const int i = e0 + lane; if (i < d.next) xl[d.nrows + i] = (R)0; }
    } else
    for (int e0 = 0; e0 < nfar; e0 += 64 * EU) {
      int32_t ec[EU]; int es[EU];
#pragma unroll
      for (int k = 0; k < EU; ++k) {
        const int i = e0 + lane + 64 * k;
        es[k] = BWD ? (i < d.npre ? i : i + f.npost) : d.npre + i;
        ec[k] = a.ext_col[i < nfar ? d.ext0 + es[k] : 0];
      }
#pragma unroll
      for (int k = 0; k < EU; ++k) pin(ec[k]);
      R xe[EU];
#pragma unroll
      for (int k = 0; k < EU; ++k) xe[k] = xc[ec[k]];
#pragma unroll
      for (int k = 0; k < EU; ++k) pin(xe[k]);
#pragma unroll
      for (int k = 0; k < EU; ++k) { const int i = e0 + lane + 64 * k; if (i < nfar) xl[d.nrows + es[k]] = xe[k]; }
    }
    if (lane == 0) { xl[d.nrows + d.next] = (R)0; ctl[0] = 0u; ctl[1] = 0u; ctl[2] = 0u; ctl[3] = 0u; }
    // (the fetcher's whole life is this branch — its own barrier, its own return: no path of the control-flow graph leads from the
    // walkers' hand-issued loads into it, which is what tools/flow_asm_linear.py can then verify)
    fetcher_barrier();
    // ---- the fetcher: near-side values out of their mailboxes in the order the walk needs them (gs_flow.hpp, one column) ----
#ifdef BW_RELAY_FETCH_PRIO
    __builtin_amdgcn_s_setprio(BW_RELAY_FETCH_PRIO);   // (measurement: the fetcher's few instructions ahead of the walkers' many on a shared SIMD)
#endif
    constexpr int U = 3;
    // (a row-sharded operator swept in a pipeline across the ranks: the extended lists — the near side's halo columns among them,
    // polled in the neighbouring rank's mailboxes)
    int nf = BWD ? f.npost : d.npre;
    int f0 = d.ext0 + (BWD ? d.npre : 0);
    if (a.xlist) {
      const int l0 = a.xlist[4 * (int64_t)ob], lf = a.xlist[4 * (int64_t)ob + 1], lb = a.xlist[4 * (int64_t)ob + 2];
      nf = __builtin_amdgcn_readfirstlane(BWD ? lb : lf);
      f0 = __builtin_amdgcn_readfirstlane(BWD ? l0 + lf : l0);
    }
    const __amdgpu_buffer_rsrc_t rs_mail = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.mbox), 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_rmail = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<void*>(a.rmbox ? a.rmbox : (const void*)a.mbox)), 0, 0x7ffffff0, 0x00020000);
    unsigned total_spins = 0;
    bool fail = false;
#ifdef BW_RELAY_STAMPS
    long long* t_seen = a.tim + 4 * (int64_t)a.nblocks + 384 * (int64_t)a.nblocks + ((int64_t)a.nmail + 1024);
    long long poll_sum = 0, poll_max = 0, npoll = 0;
#endif
#if BW_RELAY_FETCH_PD > 1 && !defined(BW_RELAY_STAMPS)
    // PIPELINED polls (round 6): a poll is a coherent load served by the memory side — ~1-2.5 us under the sweep's own traffic — and a
    // loop that waits for one poll before it issues the next sees a value, on average, one and a half round trips after it was
    // published (measured: 2.5 us at the median, profiles/r05_flow_phases.log).  Here BW_RELAY_FETCH_PD polls of the same cells are in
    // flight, issued BW_RELAY_FETCH_GAP x 64 cycles apart: a published value is seen one round trip + half a gap later.  A cell
    // that arrived is never polled again; a later poll of a lane that an earlier one already served is dropped (same value).
    constexpr int PD = BW_RELAY_FETCH_PD;
    for (int w0 = 0; w0 < nf && !fail; w0 += 64 * U) {
      int32_t mb[U]; int slot[U]; bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = w0 + 64 * u + lane;
        mb[u] = a.fl_mb[e < nf ? f0 + e : f0];
        slot[u] = a.fl_slot[e < nf ? f0 + e : f0];
        ok[u] = !(e < nf);
      }
      int first = 0;
      typename M::cell q[PD][U];
      bool is[PD][U];
      auto issue = [&](auto rc) {
        constexpr int r = decltype(rc)::value;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          is[r][u] = !ok[u] && 64 * u <= first + 64;
          if (is[r][u]) {
            if (mb[u] < 0) q[r][u] = M::load_sys(rs_rmail, (unsigned)(mb[u] & 0x7fffffff) * (unsigned)M::kBytes);   // (kRemoteCell)
            else q[r][u] = M::load(rs_mail, (unsigned)mb[u] * (unsigned)M::kBytes);
          }
        }
      };
      auto process = [&](auto rc) -> bool {
        constexpr int r = decltype(rc)::value;
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (is[r][u] && !ok[u] && M::valid(q[r][u], epoch)) { xl[slot[u]] = M::value(q[r][u]); ok[u] = true; }
        int nfirst = 64 * U;
#pragma unroll
        for (int u = U - 1; u >= 0; --u) {
          const unsigned long long m = __builtin_amdgcn_ballot_w64(!ok[u]);
          if (m) nfirst = 64 * u + (int)__builtin_ctzll(m);
        }
        if (nfirst != first && lane == 0) __hip_atomic_store(ctl, (unsigned)(w0 + nfirst < nf ? w0 + nfirst : nf), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        first = nfirst;
        return nfirst == 64 * U;
      };
      issue(std::integral_constant<int, 0>());
      if constexpr (PD > 1) { __builtin_amdgcn_s_sleep(BW_RELAY_FETCH_GAP); issue(std::integral_constant<int, 1>()); }
      if constexpr (PD > 2) { __builtin_amdgcn_s_sleep(BW_RELAY_FETCH_GAP); issue(std::integral_constant<int, 2>()); }
      if constexpr (PD > 3) { __builtin_amdgcn_s_sleep(BW_RELAY_FETCH_GAP); issue(std::integral_constant<int, 3>()); }
      bool done = false;
      while (!done && !fail) {
        if (process(std::integral_constant<int, 0>())) { done = true; break; }
        issue(std::integral_constant<int, 0>());
        if constexpr (PD > 1) { __builtin_amdgcn_s_sleep(BW_RELAY_FETCH_GAP); if (process(std::integral_constant<int, 1>())) { done = true; break; } issue(std::integral_constant<int, 1>()); }
        if constexpr (PD > 2) { __builtin_amdgcn_s_sleep(BW_RELAY_FETCH_GAP); if (process(std::integral_constant<int, 2>())) { done = true; break; } issue(std::integral_constant<int, 2>()); }
        if constexpr (PD > 3) { __builtin_amdgcn_s_sleep(BW_RELAY_FETCH_GAP); if (process(std::integral_constant<int, 3>())) { done = true; break; } issue(std::integral_constant<int, 3>()); }
        __builtin_amdgcn_s_sleep(BW_RELAY_FETCH_GAP);
        total_spins += PD;
        if (total_spins > spin_limit) fail = true;
      }
    }
#else
    for (int w0 = 0; w0 < nf && !fail; w0 += 64 * U) {
      int32_t mb[U]; int slot[U]; bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = w0 + 64 * u + lane;
        mb[u] = a.fl_mb[e < nf ? f0 + e : f0];
        slot[u] = a.fl_slot[e < nf ? f0 + e : f0];
        ok[u] = !(e < nf);
      }
      int first = 0;
      for (;;) {
        typename M::cell cl[U];
#ifdef BW_RELAY_STAMPS
        const long long tp0 = wall_clock64();
#endif
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (!ok[u] && 64 * u <= first + 64) {
            if (mb[u] < 0) cl[u] = M::load_sys(rs_rmail, (unsigned)(mb[u] & 0x7fffffff) * (unsigned)M::kBytes);   // (kRemoteCell)
            else cl[u] = M::load(rs_mail, (unsigned)mb[u] * (unsigned)M::kBytes);
          }
#ifdef BW_RELAY_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long tp1 = wall_clock64();
        poll_sum += tp1 - tp0; poll_max = poll_max > tp1 - tp0 ? poll_max : tp1 - tp0; ++npoll;
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (a.tim && mb[u] >= 0 && !ok[u] && 64 * u <= first + 64 && M::valid(cl[u], epoch)) t_seen[mb[u]] = tp1;
#endif
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (!ok[u] && 64 * u <= first + 64 && M::valid(cl[u], epoch)) { xl[slot[u]] = M::value(cl[u]); ok[u] = true; }
        int nfirst = 64 * U;
#pragma unroll
        for (int u = U - 1; u >= 0; --u) {
          const unsigned long long m = __builtin_amdgcn_ballot_w64(!ok[u]);
          if (m) nfirst = 64 * u + (int)__builtin_ctzll(m);
        }
        const bool moved = nfirst != first;
        if (moved && lane == 0) __hip_atomic_store(ctl, (unsigned)(w0 + nfirst < nf ? w0 + nfirst : nf), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        first = nfirst;
        if (nfirst == 64 * U) break;
#ifndef BW_RELAY_FETCH_NOSLEEP
        if (!moved) __builtin_amdgcn_s_sleep(BW_RELAY_FETCH_SLEEP);
#endif
        if (++total_spins > spin_limit) { fail = true; break; }
      }
    }
#endif
    if (fail && lane == 0) {
      *a.err = 1;
      __hip_atomic_store(ctl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#ifdef BW_RELAY_STAMPS
    if (a.tim && lane == 0) {
      long long* q = a.tim + 4 * (int64_t)a.nblocks + 384 * (int64_t)a.nblocks + 2 * ((int64_t)a.nmail + 1024) + 4 * (int64_t)ob;
      q[0] = npoll; q[1] = poll_sum; q[2] = poll_max; q[3] = 0;
    }
#endif
    return;
    }
  // ---- the walkers ----
  // my steps: wv, wv + W, ...; step words by scalar loads one issue ahead (gs_flow.hpp)
  const uint32_t* ax = a.aux + f.aux + (BWD ? ns + 1 : 0);
  typedef const uint32_t __attribute__((address_space(4))) cu32;
  const cu32* axc = (const cu32*)(unsigned long long)uniform_ptr(ax);
  unsigned sw_next = axc[wv < ns ? wv : ns];
  const i32x4 rs_rec = DICT ? make_rsrc(a.crec + (size_t)(uint32_t)d.row0 * (size_t)(16 * O::NCC)) : make_rsrc(a.srec + (size_t)(uint32_t)d.rec * 16);
  const i32x4 rs_b = make_rsrc(a.b + d.row0);
  const i32x4 rs_x = make_rsrc(x + d.row0);
  const i32x4 rs_mst = make_rsrc(a.mbox);
  // operands of walking step kk (one of mine) into o; steps behind the last one: a harmless re-read of the block's first chunk
  auto issue = [&](O& o, int kk) {
    const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)sw_next);
    sw_next = axc[kk + W < ns ? kk + W : ns];
    const int r0 = (int)(w & ((1u << kStepRowBits) - 1)), nr = (int)((w >> kStepRowBits) & ((1u << kStepCntBits) - 1));
    o.need = (int)(w >> (kStepRowBits + kStepCntBits));
    const int tt = lane < nr ? lane : 0;
    o.p = r0 + tt;
    const unsigned voff = (unsigned)tt * 16u;
    const unsigned s0 = (unsigned)r0 * (16u * O::KCH), sd = (unsigned)nr * 16u;
    if constexpr (DICT) {
#pragma unroll
      for (int c = 0; c < O::NCC - 1; ++c) asm_load<4>(o.cf[c], voff, rs_rec, s0 + (unsigned)c * sd);
      asm_load<O::CTAIL + 1>(o.ct, voff, rs_rec, s0 + (unsigned)(O::NCC - 1) * sd);
    } else {
#pragma unroll
      for (int c = 0; c < O::NVC - 1; ++c) asm_load<4>(o.vf[c], voff, rs_rec, s0 + (unsigned)c * sd);
      asm_load<O::VTAIL>(o.vt, voff, rs_rec, s0 + (unsigned)(O::NVC - 1) * sd);
#pragma unroll
      for (int c = 0; c < O::NCC - 1; ++c) asm_load<4>(o.cf[c], voff, rs_rec, s0 + (unsigned)(O::NVC + c) * sd);
      asm_load<O::CTAIL + 1>(o.ct, voff, rs_rec, s0 + (unsigned)(O::NVC + O::NCC - 1) * sd);
    }
    asm_load<RB / 4>(o.bb, (unsigned)tt * RB, rs_b, (unsigned)r0 * RB);
  };
  O ops[D];
  {
    // the walkers' half of the load phase: the block's own rows into LDS, a share each
    constexpr int XW = W >= 4 ? 2 : 4;
    if (a.xzero) {
      for (int p = wv * 64 + lane; p < d.nrows; p += 64 * W) xl[p] = (R)0;
    } else
    for (int p0 = wv * 64 * XW; p0 < d.nrows; p0 += 64 * XW * W) {
      R xv[XW];
#pragma unroll
      for (int k = 0; k < XW; ++k) { const int p = p0 + lane + 64 * k; xv[k] = x[d.row0 + (p < d.nrows ? p : 0)]; }
#pragma unroll
      for (int k = 0; k < XW; ++k) pin(xv[k]);
#pragma unroll
      for (int k = 0; k < XW; ++k) { const int p = p0 + lane + 64 * k; if (p < d.nrows) xl[p] = xv[k]; }
    }
    if constexpr (DICT) {   // the block's dictionary (its distinct value rows) behind the control words
      const int32_t de = a.dict_ent[ob];
      const int nchunk = (((unsigned)de >> 24) + 1) * NVC;
      const u32x4* dg = (const u32x4*)(a.dict + (size_t)(de & 0xffffff) * 16);
      u32x4* dl = (u32x4*)(lds + nxb + 16);
      for (int i = wv * 64 + lane; i < nchunk; i += 64 * W) dl[i] = dg[i];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (nothing of the compiler's in flight when the counted loads start)
#pragma unroll
    for (int j = 0; j < D; ++j) issue(ops[j], wv + j * W);
  }
  __syncthreads();
  // ---- a walker ----
  __builtin_amdgcn_s_setprio(2);
  const long long t_loaded = a.tim ? wall_clock64() : 0;
  const int32_t pubdir = BWD ? kPubBwd : kPubFwd;
  const bool mute = a.skip_pub == tk;
  const unsigned spare = (unsigned)a.nmail + (unsigned)(ob & 1023);
  bool gave_up = false, bail = false;
  const unsigned xl_base = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr(xl));
  const unsigned ctl_off = xl_base + (unsigned)nxb;   // (LDS address of the control words)
  const unsigned dict_base = ctl_off + 16u;            // (the dictionary layout: the block's distinct value rows)
#ifdef BW_RELAY_STAMPS   // (measurement builds: three wall-clock stamps per step in LDS behind the control words, dumped at the end;
  // per mailbox the time its row was published (a counted third store of the step) and the time its reader's fetcher saw it)
  const unsigned stamp_base = xl_base + (unsigned)nxb + 16u;
  i32x4 rs_pub = make_rsrc(a.tim + 4 * (int64_t)a.nblocks + 384 * (int64_t)a.nblocks);
  if (!a.tim) rs_pub.z = 0;   // (no stamps asked for: zero records, the stores fall out of range and are dropped)
#endif
  auto step = [&](const O& o, int kk) {
#ifdef BW_RELAY_STAMPS
    const long long t_rec = wall_clock64();   // the step's record has landed
#endif
    // ---- off the critical path: everything the record alone determines ----
    // (rows of more than 18 entries: the addresses are made behind the hand-over, batch by batch — D sets of such rows leave
    // no registers for them, and a compiler short of registers moves sets that are still in flight)
    constexpr bool PRE = MAXK <= 18;
    constexpr int NOFF = (PRE && !LATE) ? MAXK : 1;   // (LATE keeps the near half's addresses only: offl below)
    auto col_addr = [&](int k) {
      const u32x4 cw = o.cchunk(k >> 3);
      const unsigned w = ((k >> 1) & 3) == 0 ? cw.x : ((k >> 1) & 3) == 1 ? cw.y : ((k >> 1) & 3) == 2 ? cw.z : cw.w;
      return xl_base + ((k & 1) ? (w >> 16) : (w & 0xffffu));
    };
    unsigned off[NOFF];
#pragma unroll
    for (int k = 0; k < NOFF; ++k) off[k] = col_addr(k);
    const int32_t pw = o.pub();
    unsigned cell = ((pw & pubdir) && !mute) ? (unsigned)f.pad0 + (((unsigned)pw >> (BWD ? 11 : 0)) & 0x7ffu) : spare;
    cell *= (unsigned)M::kBytes;
    u32x4 vv[NVC];   // (the dictionary layout: the row's values out of LDS, ahead of the hand-over)
    unsigned da = 0u;
    if constexpr (DICT) {
      da = dict_base + (((unsigned)pw >> kDictIdxShift) & 0xffu) * (unsigned)(16 * NVC);
#pragma unroll
      for (int c = 0; c < NVC; ++c) vv[c] = lds_get<u32x4>(da + 16u * (unsigned)c);
#pragma unroll
      for (int c = 0; c < NVC; ++c) asm volatile("" : "+v"(vv[c]));
    }
    auto val = [&](int k) -> R { if constexpr (DICT) return value_of<R>(vv, k); else return chunk_value<R>(o, k); };
    unsigned xoff = (unsigned)o.p * (unsigned)RB;   // (the row's x: byte offset in the block, in LDS and in memory alike)
    unsigned xadr = xl_base + xoff;
    // (LATE) everything the far side and the record determine: the far half of the row sum, the row's own old x, and — so that
    // the tail is nothing but the near half's multiply-adds — the row brought to the form x_i = q0 - sum_near nv[k] x[k]:
    //   Gauss-Seidel: q0 = (b - early) rc, nv[k] = v[k] rc (rc = RN(1 / d) from the record);
    //   SOR:          q0 = (1 - omega) x_i + (omega / d)(b - early), nv[k] = v[k] (omega / d).
    // Rows whose record says "divide" (rc = 0: a diagonal outside [1e-100, 1e100]) and rows with a zero diagonal (the row keeps
    // its value, smoother.jl:87) are told apart here too; the tail only selects.
    constexpr int H = MAXK / 2, E0 = BWD ? 0 : H, L0 = BWD ? H : 0;
    R q0 = (R)0, xo_early = (R)0, early = (R)0, bb_early = (R)0;
    R nv[LATE ? H : 1];
    unsigned offl[LATE ? H : 1];
    bool keep_row = false, slow_row = false;
    unsigned long long slow_any = 0ull;
    if constexpr (LATE) {
      static_assert(!LATE || MAXK % 2 == 0, "the split sum halves the slots");
#pragma unroll
      for (int k = 0; k < H; ++k) offl[k] = col_addr(L0 + k);
      xo_early = lds_get<R>(xadr);
      constexpr int GE = 12;   // far-side gathers per batch (rows of 36 entries: 18 gathers in flight at once would cost their registers)
#pragma unroll
      for (int k0 = 0; k0 < H; k0 += GE) {
        R xe[GE];
#pragma unroll
        for (int k = 0; k < GE; ++k) if (k0 + k < H) xe[k] = lds_get<R>(col_addr(E0 + k0 + k));
#pragma unroll
        for (int k = 0; k < GE; ++k) if (k0 + k < H) early = __builtin_fma(val(E0 + k0 + k), xe[k], early);
      }
      const R dg = val(MAXK), rc = val(MAXK + 1);
      if constexpr (sizeof(R) == 8) bb_early = __hiloint2double((int)o.bb.y, (int)o.bb.x); else bb_early = __uint_as_float(o.bb);
      keep_row = dg == (R)0;
      if constexpr (SOR) {
        const R w = a.omega / dg;
        q0 = __builtin_fma(w, bb_early - early, ((R)1 - a.omega) * xo_early);
#pragma unroll
        for (int k = 0; k < H; ++k) nv[k] = -(val(L0 + k) * w);
      } else {
        slow_row = rc == (R)0 && !keep_row;
        q0 = (bb_early - early) * rc;
#pragma unroll
        for (int k = 0; k < H; ++k) nv[k] = -(val(L0 + k) * rc);
      }
      slow_any = __builtin_amdgcn_ballot_w64(slow_row);
      asm volatile("" : "+v"(q0), "+v"(xo_early));
#pragma unroll
      for (int k = 0; k < H; ++k) asm volatile("" : "+v"(nv[k]), "+v"(offl[k]));
    } else {
#pragma unroll
      for (int k = 0; k < NOFF; ++k) asm volatile("" : "+v"(off[k]));
    }
    asm volatile("" : "+v"(cell), "+v"(xoff), "+v"(xadr));
    // ---- the hand-over: every step before kk is finished and the near-side values this step reads are in LDS ----
    if (!bail) {
      // (a sibling's step arrives within a few polls: spin; a predecessor BLOCK's values may take the rest of the sweep: after
      // a few dozen polls the wave sleeps between them and gives its issue slots to the waves that have work)
      unsigned spins = 0;
      for (;;) {
        u32x4 cw;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(cw) : "v"(ctl_off) : "memory");
#ifdef BW_RELAY_POLL_V2
        // (every lane holds the same words: compared in the vector unit, one ballot for the branch)
        const bool okv = ((cw.x >= (unsigned)o.need) & (cw.y >= (unsigned)kk)) | (cw.z != 0u);
        if (__builtin_amdgcn_ballot_w64(okv) != 0ull) { bail = (unsigned)__builtin_amdgcn_readfirstlane((int)cw.z) != 0u; break; }
#else
        const unsigned pr = (unsigned)__builtin_amdgcn_readfirstlane((int)cw.x), dn = (unsigned)__builtin_amdgcn_readfirstlane((int)cw.y);
        const unsigned bl = (unsigned)__builtin_amdgcn_readfirstlane((int)cw.z);
        if (((pr >= (unsigned)o.need) & (dn >= (unsigned)kk)) | (bl != 0u)) { bail = bl != 0u; break; }
#endif
        if (++spins > BW_RELAY_SPIN) {
#ifndef BW_RELAY_NOSLEEP
          if (spins == BW_RELAY_SPIN + 1u) __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_s_sleep(1);
#endif
          if (spins > 4u * spin_limit) { gave_up = true; bail = true; break; }
        }
      }
#ifndef BW_RELAY_NOSLEEP
      if (spins > BW_RELAY_SPIN) __builtin_amdgcn_s_setprio(2);
#endif
    }
#ifdef BW_RELAY_STAMPS
    const long long t_go = wall_clock64();    // the hand-over has arrived
#endif
    asm volatile("" ::: "memory");   // the gathers stay below the hand-over
    // ---- the dependent tail ----
    R acc = (R)0;
    R xo;
    R q_late = (R)0;
    if constexpr (LATE) {
      R xv[H];
#pragma unroll
      for (int k = 0; k < H; ++k) xv[k] = lds_get<R>(offl[k]);
      xo = xo_early;
      // two chains for the long rows: the dependent latency of H multiply-adds halves, one add joins them
      R qa = q0, qb = (R)0;
#pragma unroll
      for (int k = 0; k < H; ++k) {
        if (H >= 6 && (k & 1)) qb = __builtin_fma(nv[k], xv[k], qb);
        else qa = __builtin_fma(nv[k], xv[k], qa);
      }
      q_late = H >= 6 ? qa + qb : qa;
      if (slow_any != 0ull) {   // (wave-uniform, decided above the hand-over) rows that must divide: the plain sum and the division
        asm volatile("; LATE: rows whose record asks for the division" ::: "memory");
        // (cold: the values come back out of the dictionary / the record — nothing of them is kept alive across the hand-over for this)
        auto val2 = [&](int k) -> R { if constexpr (DICT) return lds_get<R>(da + (unsigned)(RB * k)); else return chunk_value<R>(o, k); };
        R s2 = early;
#pragma unroll
        for (int k = 0; k < H; ++k) s2 += val2(L0 + k) * xv[k];
        if (slow_row) q_late = (bb_early - s2) / val2(MAXK);
      }
    } else if constexpr (PRE) {
      R xv[MAXK];
#pragma unroll
      for (int k = 0; k < MAXK; ++k) xv[k] = lds_get<R>(off[k]);
      xo = lds_get<R>(xadr);
#pragma unroll
      for (int k = 0; k < MAXK; ++k) acc += val(k) * xv[k];
    } else {
      constexpr int GB = 12;   // gathers per batch
      xo = lds_get<R>(xadr);
#pragma unroll
      for (int k0 = 0; k0 < MAXK; k0 += GB) {
        R xv[GB];
#pragma unroll
        for (int k = 0; k < GB; ++k) if (k0 + k < MAXK) xv[k] = lds_get<R>(col_addr(k0 + k));
#pragma unroll
        for (int k = 0; k < GB; ++k) if (k0 + k < MAXK) pin(xv[k]);
#pragma unroll
        for (int k = 0; k < GB; ++k) if (k0 + k < MAXK) acc += val(k0 + k) * xv[k];
      }
    }
    R q;
    if constexpr (LATE) q = keep_row ? xo : q_late;
    else {
    const R dg = val(MAXK), rc = val(MAXK + 1);
    R bbv;
    if constexpr (sizeof(R) == 8) bbv = __hiloint2double((int)o.bb.y, (int)o.bb.x); else bbv = __uint_as_float(o.bb);
    const R nn = bbv - acc;
    if (SOR) q = ((R)1 - a.omega) * xo + (a.omega / dg) * nn;
    else {
      // (bb - acc) / dg from rc = RN(1 / dg): the correctly rounded quotient (Markstein) inside the normal range — packed_row
      q = nn * rc;
      const R rem = __builtin_fma(-dg, q, nn);
      q = __builtin_fma(rem, rc, q);
      const R an = __builtin_fabs(nn);
      const bool safe = sizeof(R) == 8 ? (an > (R)1e-200 && an < (R)1e200) : (an > (R)1e-25 && an < (R)1e25);
      if (__builtin_amdgcn_ballot_w64(!(rc != (R)0 && safe) && dg != (R)0) != 0) {
        asm volatile("; rows outside the normal range: the division itself" ::: "memory");
        if (!(rc != (R)0 && safe)) q = nn / dg;
      }
    }
    q = dg != (R)0 ? q : xo;   // a zero diagonal: the row keeps its value (smoother.jl:87) — and publishes it
    }
    lds_put<R>(xadr, q);
    // (LDS operations of one wave execute in program order: the values are in place before the word moves)
    __hip_atomic_store(ctl + 1, (unsigned)(kk + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (gave_up) __hip_atomic_store(ctl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef BW_RELAY_STAMPS
    const long long t_done = wall_clock64();   // x and the word are on their way
    if (a.tim && kk < 128) {
      lds_put<long long>(stamp_base + 24u * (unsigned)kk, t_rec);
      lds_put<long long>(stamp_base + 24u * (unsigned)kk + 8u, t_go);
      lds_put<long long>(stamp_base + 24u * (unsigned)kk + 16u, t_done);
    }
    { u32x2 tv2 = {(unsigned)t_done, (unsigned)(t_done >> 32)}; asm_store(tv2, (cell / (unsigned)M::kBytes) * 8u, rs_pub); }
#endif
    // ---- behind the hand-over: the row into its mailbox (tagged with the sweep's epoch) and into x ----
    if constexpr (sizeof(R) == 8) {
      const unsigned qlo = (unsigned)__double2loint(q), qhi = (unsigned)__double2hiint(q);
      u32x4 cv = {qlo, epoch, qhi, epoch};
      asm_store_sc1(cv, cell, rs_mst);
      u32x2 xv2 = {qlo, qhi};
      asm_store(xv2, xoff, rs_x);
    } else {
      u32x2 cv = {__float_as_uint(q), epoch};
      asm_store_sc1(cv, cell, rs_mst);
      asm_store(__float_as_uint(q), xoff, rs_x);
    }
#ifdef BW_FLOW_STEP_STAMPS
    if (a.tim && lane == 0) a.tim[4 * (int64_t)a.nblocks + 128 * (int64_t)ob + kk] = wall_clock64();
#endif
  };
  // The pipeline of gs_flow.hpp over MY steps (j-th one: wv + j W): D register sets, vmcnt counted exactly per wave
#ifdef BW_RELAY_STAMPS
  constexpr int L = O::NLOAD, S = 3;
#else
  constexpr int L = O::NLOAD, S = 2;
#endif
  const int nsw = ns > wv ? (ns - wv + W - 1) / W : 0;
  int k = 0;
  if (nsw >= D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {   // the first round: set j was followed by D - 1 - j more sets and j whole steps
      if (j == 0) flow_wait<(D - 1) * L>(ops[0]);
      else if (j == 1) flow_wait<(D - 2) * L + 1 * (L + S)>(ops[1 < D ? 1 : 0]);
      else if (j == 2) flow_wait<(D > 2 ? D - 3 : 0) * L + 2 * (L + S)>(ops[2 < D ? 2 : 0]);
      else flow_wait<(D > 3 ? D - 4 : 0) * L + 3 * (L + S)>(ops[3 < D ? 3 : 0]);
      step(ops[j], wv + j * W);
      issue(ops[j], wv + (j + D) * W);
    }
    for (k = D; k + D <= nsw; k += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        flow_wait<(D - 1) * (L + S)>(ops[j]);
        step(ops[j], wv + (k + j) * W);
        issue(ops[j], wv + (k + j + D) * W);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < D - 1; ++j)
    if (k + j < nsw) { flow_wait<0>(ops[j]); step(ops[j], wv + (k + j) * W); }
  if (gave_up && lane == 0) *a.err = 2;
#ifdef BW_RELAY_STAMPS
  if (a.tim) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int kk = wv; kk < ns && kk < 128; kk += W)
      if (lane < 3) a.tim[4 * (int64_t)a.nblocks + 384 * (int64_t)ob + 3 * kk + lane] = lds_get<long long>(stamp_base + 24u * (unsigned)kk + 8u * (unsigned)lane);
  }
#endif
  if (a.tim && lane == 0) {
    long long* tt = a.tim + 4 * (int64_t)ob;
    if (wv == 0) {
      tt[0] = t_start; tt[1] = t_loaded;
      unsigned xcc = 0;
#ifdef BW_RELAY_STAMPS
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));   // (which XCD the block ran on: hand-offs inside / across XCDs)
#endif
      tt[3] = (long long)(xcc & 15u);
    }
    if (wv == (ns - 1) % W) tt[2] = wall_clock64();
  }
}

// PERSISTENT form: a launch of fewer workgroups than blocks keeps every workgroup resident for the whole sweep — each draws
// tickets until they run out.  That is what several sweeps sharing one device need (the ranks of a row-sharded level on a
// single GPU: a workgroup of rank p may wait for values of rank p - 1, whose workgroups must then BE resident, not queued
// behind it), and what bounds the resident workgroups of a launch.  A launch of one workgroup per block is the plain form.
template <typename R, bool SOR, bool BWD, int MAXK, int W, bool DICT = false, bool LATE = false>
__global__ __launch_bounds__(64 * (W + 1), DICT ? RelayWavesD<MAXK>::value : RelayWaves<MAXK>::value) void gs_bw_relay_kernel(FlowArgs<R> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_all[];
  static_assert(W >= 2 && W <= 7, "walker waves per block");
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = (int)(threadIdx.x & 63u);
  const unsigned int units = (unsigned int)a.nblocks;
  const bool persistent = gridDim.x < units;
  const unsigned int draws = units + (persistent ? gridDim.x : 0u);   // (every persistent workgroup draws one ticket past the last block)
  unsigned long long* s_ticket = (unsigned long long*)lds_all;
  if (blockIdx.x < 2048u) for (unsigned i = 0; i < (blockIdx.x >> 5); ++i) __builtin_amdgcn_s_sleep(14);   // (staggered draws: gs_flow.hpp)
  for (;;) {
    const long long t_start = a.tim ? wall_clock64() : 0;
    if (threadIdx.x == 0) *s_ticket = atomicAdd(a.head, 1ull);
    __syncthreads();
    const unsigned long long tv = *s_ticket;
    const unsigned long long ticket = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(tv >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)tv);
    __syncthreads();   // (the word is x[0] from here on)
    const unsigned int ut = (unsigned int)ticket, sweeps = (unsigned int)(ticket >> 32);
    // whoever draws the launch's last ticket starts the next sweep's count (every other draw of this launch has happened by then)
    if (threadIdx.x == 0 && ut == draws - 1u)
      __hip_atomic_store(a.head, (unsigned long long)(sweeps + 2u >= 0x80000000u ? 0u : sweeps + 1u) << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ut >= units) return;
    const unsigned int epoch = a.epoch ? a.epoch : sweeps + 1u;   // (a.epoch: the tag of a sweep that several launches share, amghip_dist.hpp)
    relay_block<R, SOR, BWD, MAXK, W, DICT, LATE>(a, lds_all, wv, lane, ut, epoch, t_start);
    if (!persistent) return;
    __syncthreads();   // (everybody is done with this block's LDS)
  }
}
static_assert(RelayDepth<6>::value <= 4 && RelayDepth<12>::value <= 4 && RelayDepth<18>::value <= 4, "the first round of the relay's pipeline is written out for up to four sets");

// measurement knob: extra dynamic LDS per workgroup (bounds the blocks resident per CU)
inline size_t& relay_lds_pad() { static size_t pad = 0; return pad; }

template <typename R, bool SOR, bool BWD, int MAXK, int W, bool DICT = false, bool LATE = false>
inline hipError_t sweep_relay_launch(const FlowArgs<R>& a, size_t lds, hipStream_t st) {
  auto* fn = gs_bw_relay_kernel<R, SOR, BWD, MAXK, W, DICT, LATE>;
  if (relay_lds_pad()) {
    lds += relay_lds_pad();
    static hipError_t once = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (once != hipSuccess) return once;
    hipLaunchKernelGGL(fn, dim3((unsigned)(a.grid > 0 && a.grid < a.nblocks ? a.grid : a.nblocks)), dim3(64 * (W + 1)), lds, st, a);
    return hipGetLastError();
  }
#ifdef BW_RELAY_STAMPS
  lds += 128 * 24 + (DICT ? kDictLdsMax : 0);
#endif
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(fn, dim3((unsigned)(a.grid > 0 && a.grid < a.nblocks ? a.grid : a.nblocks)), dim3(64 * (W + 1)), lds, st, a);
  return hipGetLastError();
}
// workgroups of the relayed kernel one device keeps resident at once (what a persistent launch may ask for)
template <typename R, int MAXK, int W, bool DICT = false>
inline int relay_resident(size_t lds) {
  int per_cu = 0, dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gs_bw_relay_kernel<R, false, false, MAXK, W, DICT>, 64 * (W + 1), lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return per_cu * prop.multiProcessorCount;
}
template <typename R>
inline int relay_resident_blocks(int maxk, size_t lds, bool dict = false) {   // (dict: the kernel on the dictionary layout, lds with its dictionary)
  switch (maxk) {
    case 6: return dict ? relay_resident<R, 6, BW_RELAY_W, true>(lds) : relay_resident<R, 6, BW_RELAY_W>(lds);
    case 12: return dict ? relay_resident<R, 12, BW_RELAY_W, true>(lds) : relay_resident<R, 12, BW_RELAY_W>(lds);
#if BW_PLAN_MAXK >= 18
    case 18: return dict ? relay_resident<R, 18, BW_RELAY_W, true>(lds) : relay_resident<R, 18, BW_RELAY_W>(lds);
#endif
  }
  return 0;
}

template <typename R, int MAXK, int W, bool DICT, bool LATE>
inline hipError_t sweep_relay_d(const FlowArgs<R>& a, size_t lds, bool sor, bool backward, hipStream_t st) {
  if (sor) return backward ? sweep_relay_launch<R, true, true, MAXK, W, DICT, LATE>(a, lds, st) : sweep_relay_launch<R, true, false, MAXK, W, DICT, LATE>(a, lds, st);
  return backward ? sweep_relay_launch<R, false, true, MAXK, W, DICT, LATE>(a, lds, st) : sweep_relay_launch<R, false, false, MAXK, W, DICT, LATE>(a, lds, st);
}
template <typename R, int MAXK, int W>
inline hipError_t sweep_relay_k(const FlowArgs<R>& a, size_t lds, bool sor, bool backward, hipStream_t st) {
#ifdef BW_EXTRA_DICT
  constexpr bool kDictOk = true;      // (tools: the dictionary layout and the split sum under the extra row length too)
#else
  constexpr bool kDictOk = MAXK <= 18;
#endif
  if constexpr (kDictOk) {   // (the dictionary layout, where the schedule carries one: FlowDict; the split row sum where the records allow it: a.late)
    if (a.crec) return a.late ? sweep_relay_d<R, MAXK, W, true, true>(a, lds, sor, backward, st) : sweep_relay_d<R, MAXK, W, true, false>(a, lds, sor, backward, st);
    if constexpr (MAXK <= 18) if (a.late) return sweep_relay_d<R, MAXK, W, false, true>(a, lds, sor, backward, st);
  }
  return sweep_relay_d<R, MAXK, W, false, false>(a, lds, sor, backward, st);
}
// one column, W walker waves per block (the kernels instantiated: BW_RELAY_W, or the set a tool asks for with BW_RELAY_ALL_W)
template <typename R, int MAXK>
inline hipError_t sweep_relay_w(FlowArgs<R> a, size_t lds_max, bool sor, bool backward, int w, hipStream_t st) {
  a.ncols = 1; a.ngroups = 1; a.lds_stride = 0;
  switch (w) {
#ifdef BW_RELAY_ALL_W
    case 2: return sweep_relay_k<R, MAXK, 2>(a, lds_max, sor, backward, st);
    case 3: return sweep_relay_k<R, MAXK, 3>(a, lds_max, sor, backward, st);
    case 4: return sweep_relay_k<R, MAXK, 4>(a, lds_max, sor, backward, st);
#else
    case BW_RELAY_W: return sweep_relay_k<R, MAXK, BW_RELAY_W>(a, lds_max, sor, backward, st);
#endif
  }
  return hipErrorInvalidValue;
}
template <typename R>
inline hipError_t sweep_relay(const FlowArgs<R>& a, int maxk, size_t lds_max, bool sor, bool backward, hipStream_t st, int w = BW_RELAY_W) {
  switch (maxk) {
    case 6: return sweep_relay_w<R, 6>(a, lds_max, sor, backward, w, st);
    case 12: return sweep_relay_w<R, 12>(a, lds_max, sor, backward, w, st);
#if BW_PLAN_MAXK >= 18
    case 18: return sweep_relay_w<R, 18>(a, lds_max, sor, backward, w, st);
#endif
#ifdef BW_EXTRA_MAXK   // (tools: one more row length, e.g. the 35-entry rows of the third level of the 256^3 hierarchy)
    case BW_EXTRA_MAXK: return sweep_relay_w<R, BW_EXTRA_MAXK>(a, lds_max, sor, backward, w, st);
#endif
  }
  return hipErrorInvalidValue;
}

}  // namespace bw
}  // namespace amgh
