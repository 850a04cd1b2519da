// gs_flow.hpp — the wavefront of blocks of gs_blocks.hpp executed as a DATAFLOW: blocks start on finished faces.
// Included by gs_schedule.hpp (single-column sweeps of block-ordered levels) and tools/block_wave_bench.hip.
//
// smoother.jl:61-90 (gs!) / :193-221 (sor_step!) in exact lexicographic order, the same plan, partition and arithmetic as
// gs_blocks.hpp — what changes is WHEN a block may run.  The chained kernel there hands a block over as a whole: a successor
// waits for its predecessors' flags, so along the critical path every depth of the quotient graph costs the deepest block's
// whole walk (22 steps of an 8 x 8 x 8 block) plus a flag hand-off, and a block holds 41-124 KB of LDS (its record) from its
// load to its last step.  Here
//   * a row that another block reads is PUBLISHED the moment it is computed: one 16-byte write-through store of two
//     self-tagged 8-byte granules {value half, sweep epoch} into the row's MAILBOX (the data is the flag: no drain, no fence,
//     no separate flag word — the "R2" hand-off of the CDNA4 guide);
//   * a second wave of the block's workgroup, the FETCHER, polls the mailboxes of the block's near-side external columns in
//     the order the walk needs them, stores the values into the block's LDS x and advances a progress word in LDS; the
//     walker checks that word before each step.  A successor therefore runs a few steps — not a block — behind its
//     predecessor: 766 + fill serial steps on the 256^3 fine level instead of 94 x 22;
//   * the packed rows are STREAMED from HBM straight into the walker's registers, a few steps ahead of their use (one
//     coalesced 16-byte load per chunk: the record is stored chunk-major per step), b and the publish word with them;
//     LDS holds only x (own rows + external columns): 7-11 KB per block instead of 51-136, so 8 blocks per CU are resident
//     and the wide middle of the wavefront is no longer bound by LDS slots.
// Order of operations per row: unchanged (products in stored entry order, separately rounded, one quotient) — the iterate
// is the scalar loop's bit for bit; tests/test_gpu_flow.py compares it with the chained kernel, the host execution of the plan and the scalar loop.
//
// Progress guarantee: workgroups draw tickets in the order of the quotient DAG's depths (as the chained kernel); a block
// only ever waits for values of blocks with smaller tickets, which are running or done whatever the dispatch order.
// The data dependencies must carry the anti-dependencies as well (a block's far-side x is read at its start, before any
// successor may overwrite it): true when the pattern is structurally symmetric — plan-time check, otherwise the chained
// kernel keeps the level.  Every poll is bounded; a give-up sets *err (read by the solve path: AMGH_ESTATE).
#pragma once
#include "gs_blocks.hpp"

namespace amgh {
namespace bw {

constexpr int32_t kPubFwd = 1 << 30;    // pub word: forward mailbox | backward mailbox << 11 (relative to the block's first) | directions in which the row is published
constexpr int32_t kPubBwd = (int32_t)(1u << 31);
constexpr int32_t kPubMask = (1 << 30) - 1;
#ifndef BW_FLOW_WAVES
#define BW_FLOW_WAVES 4                 // waves per SIMD the kernel is compiled for (128 registers per lane: 8 blocks per CU)
#endif
#ifndef BW_FLOW_DEPTH
#define BW_FLOW_DEPTH 4                 // steps whose operands are in flight ahead of the walk
#endif

struct FlowDesc {      // per block, beside Desc (16 bytes)
  int32_t aux;         // offset into the aux array: step words of the forward walk [nlev + 1], then of the backward walk [nlev + 1]
  int32_t npost;       // external columns behind the block that are rows (not halo columns of a sharded operator)
  int32_t pad0, pad1;  // pad0: the block's first mailbox
};
// step word k of a walk (in walking order): first row | rows << 11 | near-side values the step needs in LDS << 18; word nlev: 0 | 0 | all
constexpr int kStepRowBits = 11, kStepCntBits = 7, kStepNeedMax = (1 << 14) - 1;
inline uint32_t step_word(int r0, int nr, int need) { return (uint32_t)r0 | ((uint32_t)nr << kStepRowBits) | ((uint32_t)need << (kStepRowBits + kStepCntBits)); }

constexpr int32_t kRemoteCell = (int32_t)(1u << 31);   // fetch-list entry: a mailbox of the NEIGHBOURING rank's array (FlowArgs::rmbox), not of this one

// A row-sharded operator (amghip_dist.hpp): rows = this rank's rows, columns [n, n + nlo) = halo entries owned by lower
// ranks, [n + nlo, ncols) by higher ranks.  For a sweep PIPELINED across the ranks the rows a higher rank reads get a
// forward mailbox (pub_f), the rows a lower rank reads a backward one (pub_b) — the neighbour's fetchers poll them
// through a peer mapping —, and this rank's blocks get fetch lists that hold their lower (forward sweep) / upper
// (backward sweep) halo columns beside the columns of predecessor blocks: the extended lists of FlowX.
struct FlowHalo {
  int64_t nlo = 0;                       // halo columns owned by lower ranks (they come first)
  const unsigned char* pub_f = nullptr;  // per natural local row: read by a higher rank
  const unsigned char* pub_b = nullptr;  // ... by a lower rank
};
struct FlowX {
  bool on = false;
  std::vector<uint32_t> aux;        // step words with the extended lists' counts (same offsets as Flow::aux)
  std::vector<int32_t> fl_mb;       // per block: forward list, then backward list; a halo entry holds kRemoteCell | halo index until the neighbour's cells are known
  std::vector<uint16_t> fl_slot;
  std::vector<int32_t> list;        // per block {first entry, forward entries, backward entries, 0}
  std::vector<int32_t> row_cell_f, row_cell_b;   // per natural local row: its forward / backward mailbox (-1: none)
};

// The DICTIONARY layout of the records (round 5): rows whose values (entries, diagonal, reciprocal: the row's value chunks,
// bit for bit) are equal share ONE copy per block — a block's distinct value rows form its dictionary (loaded into LDS with
// the block), a row's streamed record shrinks to its column chunks, the dictionary index riding in bits 22-29 of the publish
// word: 16 instead of 80 bytes per 7-point row, 48 instead of 208 per 19-point row.  Lossless and general (a constant-
// coefficient stencil has a handful of value rows, its Galerkin coarse operators a few dozen per block); a level with a block of
// more than 256 distinct rows, or a dictionary beyond kDictLdsMax, keeps the plain records.
constexpr int kDictIdxShift = 22, kDictMaxRows = 256;
constexpr size_t kDictLdsMax = 24 * 1024;
struct FlowDict {
  bool on = false;
  std::vector<unsigned char, NoInit<unsigned char>> crec;   // per row its column chunks, chunk-major per step; block ob's rows start at 16 * ncc * row0
  std::vector<unsigned char> dict;                            // the blocks' dictionaries, back to back (value chunks of each distinct row)
  std::vector<int32_t> ent;                                   // per block: offset into dict in 16-byte units | (rows - 1) << 24
  size_t lds_max = 0;                                         // x + control words + the largest dictionary
};

struct Flow {
  FlowX x;
  FlowDict dc;
  std::vector<FlowDesc> fd;
  std::vector<uint32_t> aux;
  std::vector<int32_t> fl_mb;       // per block at ext0: mailboxes of the forward fetch list (npre entries, in the order of
  std::vector<uint16_t> fl_slot;    // first use), then of the backward list (npost entries); and the LDS x slot each one fills
  std::vector<int32_t> pub;         // per row in block order
  std::vector<unsigned char, NoInit<unsigned char>> srec;   // the records, chunk-major per step (same offsets as Plan::rec)
  int64_t nmail = 0;
  size_t lds_max = 0;
};

// true when every off-diagonal entry (i, c) with c a row has its transpose (c, i) stored
inline bool structurally_symmetric(int64_t n, const int32_t* rowptr, const int32_t* col, int threads) {
  std::atomic<int> bad{0};
  parallel_for(std::max(1, threads), [&](int t, int TT) {
    for (int64_t i = n * t / TT; i < n * (t + 1) / TT && !bad.load(std::memory_order_relaxed); ++i)
      for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
        const int32_t c = col[j];
        if (c == i || c >= n) continue;
        bool found = false;
        for (int32_t q = rowptr[c]; q < rowptr[c + 1]; ++q) if (col[q] == i) { found = true; break; }
        if (!found) { bad.store(1); break; }
      }
  });
  return bad.load() == 0;
}

// The dataflow view of a plan: mailboxes, fetch lists in order of use, the records transposed for coalesced streaming.
// inplace: the plan's row-major records are turned into the chunk-major ones where they lie (F.srec stays empty; P.rec is
// then what the dataflow kernel streams, and no longer what the chained / launched kernels or the host execution read)
template <typename R>
bool flow_build(Plan& P, int threads, Flow* out, bool inplace = false, const FlowHalo* hx = nullptr, bool want_dict = true) {
  Flow& F = *out;
  F = Flow();
  const int64_t n = P.n;
  const int32_t B = (int32_t)P.blocks.size();
  typedef Packed<R> Pk;
  // rows other blocks read: in which sweep direction they read them as NEW values, and which block reads them first
  std::vector<int32_t> cell_f(n, -1), cell_b(n, -1);   // the row's mailbox in forward / backward sweeps (first: the reading block)
  for (int32_t ob = 0; ob < B; ++ob) {
    const Desc& d = P.blocks[ob];
    for (int32_t e = 0; e < d.next; ++e) {
      const int32_t q = P.ext_col[d.ext0 + e];
      if (q >= n) continue;                                 // halo column: never written
      if (e < d.npre) { if (cell_f[q] < 0 || ob < cell_f[q]) cell_f[q] = ob; }   // before the block: a new value of the forward sweep
      else cell_b[q] = std::max(cell_b[q], ob);
    }
  }
  if (hx) {   // rows a neighbouring rank reads: a reader behind every block (forward) / before every block (backward)
    for (int64_t q = 0; q < n; ++q) {
      const int32_t i = P.perm[q];
      if (hx->pub_f && hx->pub_f[i] && cell_f[q] < 0) cell_f[q] = B;
      if (hx->pub_b && hx->pub_b[i] && cell_b[q] < 0) cell_b[q] = B;
    }
  }
  // mailboxes, block by block: first the rows published forward grouped by reading block, then the rows published backward
  // likewise (a row read in both directions has one cell in each part), in block order inside a group — what one reader
  // fetches from one block is then a run of consecutive cells in the order it needs them (coalesced polls), and what a step
  // publishes towards one reader is a run as well (merged write-through stores).  The publish word of a row: its forward
  // cell | backward cell << 11 (both relative to the block's first cell) | the directions
  int64_t nm = 0;
  F.pub.assign(n, 0);
  F.fd.resize(B);
  {
    std::vector<int64_t> base(B + 1, 0);
    for (int32_t ob = 0; ob < B; ++ob) {
      const Desc& d = P.blocks[ob];
      int64_t c = 0;
      for (int32_t p = 0; p < d.nrows; ++p) c += (cell_f[d.row0 + p] >= 0) + (cell_b[d.row0 + p] >= 0);
      if (c >= (1 << 11)) return false;
      if ((((size_t)(d.nrows + d.next + 1) * sizeof(R) + 15) & ~(size_t)15) + 16 > 64 * 1024) return false;   // (the kernel's LDS: x of the block)
      base[ob + 1] = base[ob] + c;
    }
    nm = base[B];
    if (nm >= kPubMask || (nm + 1024) * (int64_t)(2 * sizeof(R)) >= 0x7ffffff0ll) return false;   // (the mailboxes: one buffer descriptor)
    parallel_for(std::max(1, threads), [&](int t, int TT) {
      std::vector<int32_t> rows;
      for (int32_t ob = B * (int64_t)t / TT; ob < B * (int64_t)(t + 1) / TT; ++ob) {
        const Desc& d = P.blocks[ob];
        F.fd[ob].pad0 = (int32_t)base[ob];   // (mail0: the block's first cell)
        int32_t local = 0;
        for (int dir = 0; dir < 2; ++dir) {
          std::vector<int32_t>& cl = dir ? cell_b : cell_f;
          rows.clear();
          for (int32_t p = 0; p < d.nrows; ++p) if (cl[d.row0 + p] >= 0) rows.push_back(d.row0 + p);
          std::stable_sort(rows.begin(), rows.end(), [&](int32_t x, int32_t y) { return cl[x] < cl[y]; });
          for (int32_t q : rows) {
            F.pub[q] |= dir ? ((local << 11) | kPubBwd) : (local | kPubFwd);
            cl[q] = (int32_t)base[ob] + local;    // (from here on: the absolute cell)
            ++local;
          }
        }
      }
    });
  }
  F.nmail = nm;
  int64_t aux_total = 0;
  for (int32_t ob = 0; ob < B; ++ob) {
    F.fd[ob].aux = (int32_t)aux_total;
    aux_total += 2 * ((int64_t)P.blocks[ob].nlev + 1);
    if (aux_total > INT32_MAX || P.blocks[ob].nrows >= (1 << kStepRowBits) || P.blocks[ob].next > kStepNeedMax) return false;
  }
  F.aux.assign((size_t)aux_total, 0);
  F.fl_mb.assign(std::max<size_t>(1, P.ext_col.size()), 0);
  F.fl_slot.assign(std::max<size_t>(1, P.ext_col.size()), 0);
  if (!inplace) F.srec.resize(P.rec.size());
  // the extended lists of a row-sharded operator (FlowX): collected per block, laid out behind the loop
  std::vector<std::vector<int32_t>> xl_mb(hx ? (size_t)B : 0);
  std::vector<std::vector<uint16_t>> xl_slot(hx ? (size_t)B : 0);
  std::vector<int32_t> xl_nf(hx ? (size_t)B : 0, 0);
  std::atomic<int> bad_x{0};
  // the dictionary layout: built for every block beside the plain record (dropped again if any block does not qualify)
  const int ncc_all = B > 0 ? Pk::ncc(P.blocks[0].maxk) : 1;
  std::vector<std::vector<unsigned char>> dict_of(want_dict ? (size_t)B : 0);
  std::atomic<int> bad_dict{0};
  std::atomic<size_t> lds_dict_max{0};
  if (!want_dict) bad_dict.store(1);
  else if (n * (int64_t)ncc_all * 16 < (int64_t)0x7ffffff0ll) F.dc.crec.resize((size_t)n * (size_t)ncc_all * 16); else bad_dict.store(1);
  if (hx) F.x.aux.assign((size_t)aux_total, 0);
  std::atomic<size_t> lds_max{0};
  parallel_for(std::max(1, threads), [&](int t, int TT) {
    std::vector<int32_t> use, order;
    std::vector<unsigned char> tmp;   // (in place: one block's rows, transposed, before they go back)
    size_t my_lds = 0;
    for (int32_t ob = B * (int64_t)t / TT; ob < B * (int64_t)(t + 1) / TT; ++ob) {
      const Desc& d = P.blocks[ob];
      FlowDesc& f = F.fd[ob];
      const unsigned char* rec = P.rec.data() + (size_t)d.rec * 16;
      const size_t rows_bytes = (size_t)d.nrows * Pk::row_bytes(d.maxk);
      if (inplace) tmp.resize(rows_bytes);
      unsigned char* srec = inplace ? tmp.data() : F.srec.data() + (size_t)d.rec * 16;
      const size_t rs = Pk::row_bytes(d.maxk);
      const int kch = Pk::chunks(d.maxk), nvc = Pk::nvc(d.maxk);
      const uint16_t* stp = (const uint16_t*)(rec + (size_t)d.nrows * rs);
      const int ns = d.nlev;
      uint32_t* ax = F.aux.data() + f.aux;
      // first / last step that reads each external slot
      use.assign((size_t)d.next * 2, 0);
      for (int32_t e = 0; e < d.next; ++e) { use[2 * e] = INT32_MAX; use[2 * e + 1] = -1; }
      for (int s = 0; s < ns; ++s)
        for (int32_t p = stp[s]; p < stp[s + 1]; ++p) {
          const uint16_t* cc = (const uint16_t*)(rec + (size_t)p * rs + (size_t)16 * nvc);
          for (int k = 0; k < d.maxk; ++k) {
            const int32_t slot = (int32_t)(cc[k] / sizeof(R));
            if (slot < d.nrows || slot >= d.nrows + d.next) continue;
            const int32_t e = slot - d.nrows;
            use[2 * e] = std::min(use[2 * e], s); use[2 * e + 1] = std::max(use[2 * e + 1], s);
          }
        }
      int32_t npost = 0;
      for (int32_t e = d.npre; e < d.next; ++e) if (P.ext_col[d.ext0 + e] < n) ++npost;
      f.npost = npost; f.pad1 = 0;
      // forward list: near side = columns before the block, by first use; need_fwd[k] = entries step k needs complete
      order.resize(d.npre);
      for (int32_t e = 0; e < d.npre; ++e) order[e] = e;
      std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return use[2 * a] < use[2 * b]; });
      {
        int32_t done = 0;
        for (int k = 0; k < ns; ++k) {
          while (done < d.npre && use[2 * order[done]] <= k) ++done;
          ax[k] = step_word(stp[k], stp[k + 1] - stp[k], done);
        }
        ax[ns] = step_word(0, 0, d.npre);
        for (int32_t i = 0; i < d.npre; ++i) {
          const int32_t e = order[i];
          F.fl_mb[d.ext0 + i] = cell_f[P.ext_col[d.ext0 + e]];
          F.fl_slot[d.ext0 + i] = (uint16_t)(d.nrows + e);
        }
      }
      // backward list: columns behind the block (rows only), by first use in the reversed walk (step ns - 1 - k)
      order.resize(npost);
      for (int32_t i = 0; i < npost; ++i) order[i] = d.npre + i;    // (halo columns sort last in ext_col: the first npost are rows)
      std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return use[2 * a + 1] > use[2 * b + 1]; });
      {
        int32_t done = 0;
        for (int k = 0; k < ns; ++k) {
          while (done < npost && ns - 1 - use[2 * order[done] + 1] <= k) ++done;
          ax[ns + 1 + k] = step_word(stp[ns - 1 - k], stp[ns - k] - stp[ns - 1 - k], done);
        }
        ax[2 * ns + 1] = step_word(0, 0, npost);
        for (int32_t i = 0; i < npost; ++i) {
          const int32_t e = order[i];
          F.fl_mb[d.ext0 + d.npre + i] = cell_b[P.ext_col[d.ext0 + e]];
          F.fl_slot[d.ext0 + d.npre + i] = (uint16_t)(d.nrows + e);
        }
      }
      if (hx) {
        // the same two lists with the halo columns of the near side in them: lower ranks' entries are new values of a forward
        // sweep (they lie before every row of this rank), higher ranks' entries of a backward one
        uint32_t* axx = F.x.aux.data() + f.aux;
        std::vector<int32_t>& lm = xl_mb[(size_t)ob];
        std::vector<uint16_t>& ls = xl_slot[(size_t)ob];
        auto entry_mb = [&](int32_t e, bool fwd) -> int32_t {
          const int32_t q = P.ext_col[d.ext0 + e];
          if (q >= n) return kRemoteCell | (int32_t)(q - n);
          return fwd ? cell_f[q] : cell_b[q];
        };
        order.clear();
        for (int32_t e = 0; e < d.npre; ++e) order.push_back(e);
        for (int32_t e = d.npre + npost; e < d.next; ++e) if (P.ext_col[d.ext0 + e] < n + hx->nlo && use[2 * e + 1] >= 0) order.push_back(e);
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return use[2 * a] < use[2 * b]; });
        {
          int32_t done = 0;
          const int32_t cnt = (int32_t)order.size();
          for (int k = 0; k < ns; ++k) {
            while (done < cnt && use[2 * order[done]] <= k) ++done;
            axx[k] = step_word(stp[k], stp[k + 1] - stp[k], done);
          }
          axx[ns] = step_word(0, 0, cnt);
          for (int32_t e : order) { lm.push_back(entry_mb(e, true)); ls.push_back((uint16_t)(d.nrows + e)); }
          xl_nf[(size_t)ob] = cnt;
        }
        order.clear();
        for (int32_t i = 0; i < npost; ++i) order.push_back(d.npre + i);
        for (int32_t e = d.npre + npost; e < d.next; ++e) if (P.ext_col[d.ext0 + e] >= n + hx->nlo && use[2 * e + 1] >= 0) order.push_back(e);
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return use[2 * a + 1] > use[2 * b + 1]; });
        {
          int32_t done = 0;
          const int32_t cnt = (int32_t)order.size();
          for (int k = 0; k < ns; ++k) {
            while (done < cnt && ns - 1 - use[2 * order[done] + 1] <= k) ++done;
            axx[ns + 1 + k] = step_word(stp[ns - 1 - k], stp[ns - k] - stp[ns - 1 - k], done);
          }
          axx[2 * ns + 1] = step_word(0, 0, cnt);
          for (int32_t e : order) { lm.push_back(entry_mb(e, false)); ls.push_back((uint16_t)(d.nrows + e)); }
        }
        if (lm.size() > (size_t)kStepNeedMax) bad_x.store(1);
      }
      if (bad_dict.load(std::memory_order_relaxed) == 0) {
        // distinct value rows of the block (raw bytes of the value chunks), in order of first appearance
        const size_t vb = (size_t)16 * nvc;
        const int ncc = Pk::ncc(d.maxk);
        std::vector<unsigned char>& dv = dict_of[(size_t)ob];
        std::vector<int32_t> idx_of((size_t)d.nrows);
        {
          std::vector<std::pair<uint64_t, int32_t>> seen;   // (hash, dictionary row): a block has few distinct rows — a short list
          for (int32_t pr = 0; pr < d.nrows; ++pr) {
            const unsigned char* vrow = rec + (size_t)pr * rs;
            uint64_t h = 1469598103934665603ull;
            for (size_t q = 0; q < vb; q += 8) { uint64_t w; std::memcpy(&w, vrow + q, 8); h = (h ^ w) * 1099511628211ull; }
            int32_t found = -1;
            for (const auto& e : seen)
              if (e.first == h && std::memcmp(dv.data() + (size_t)e.second * vb, vrow, vb) == 0) { found = e.second; break; }
            if (found < 0) {
              found = (int32_t)(dv.size() / vb);
              if (found >= kDictMaxRows) { bad_dict.store(1); break; }
              dv.insert(dv.end(), vrow, vrow + vb);
              seen.push_back({h, found});
            }
            idx_of[(size_t)pr] = found;
          }
        }
        if (bad_dict.load(std::memory_order_relaxed) == 0) {
          if (dv.size() > kDictLdsMax) bad_dict.store(1);
          unsigned char* crec = F.dc.crec.data() + (size_t)d.row0 * (size_t)ncc * 16;
          const int cdw = ((d.maxk + 1) / 2 - 1) % 4 + 1;
          for (int sx = 0; sx < ns; ++sx) {
            const int32_t r0 = stp[sx], nr = stp[sx + 1] - stp[sx];
            for (int c = 0; c < ncc; ++c)
              for (int32_t q = 0; q < nr; ++q) {
                unsigned char* dst = crec + ((size_t)ncc * r0 + (size_t)c * nr + q) * 16;
                std::memcpy(dst, rec + (size_t)(r0 + q) * rs + (size_t)16 * (nvc + c), 16);
                if (c == ncc - 1) {
                  const int32_t pw = F.pub[d.row0 + r0 + q] | (idx_of[(size_t)(r0 + q)] << kDictIdxShift);
                  std::memcpy(dst + 4 * cdw, &pw, 4);
                }
              }
          }
          size_t cur = lds_dict_max.load();
          while (dv.size() > cur && !lds_dict_max.compare_exchange_weak(cur, dv.size())) {}
        }
      }
      // the record, chunk-major per step: chunk c of the step's row t at 16 (kch stp[s] + c nr + t)
      for (int s = 0; s < ns; ++s) {
        const int32_t r0 = stp[s], nr = stp[s + 1] - stp[s];
        for (int c = 0; c < kch; ++c)
          for (int32_t q = 0; q < nr; ++q)
            std::memcpy(srec + ((size_t)kch * r0 + (size_t)c * nr + q) * 16, rec + (size_t)(r0 + q) * rs + (size_t)16 * c, 16);
      }
      // the publish word rides in the spare dword of the row's last column chunk
      {
        const int cdw = ((d.maxk + 1) / 2 - 1) % 4 + 1;   // dwords of that chunk that hold columns (FlowOps::CTAIL)
        for (int s = 0; s < ns; ++s) {
          const int32_t r0 = stp[s], nr = stp[s + 1] - stp[s];
          for (int32_t q = 0; q < nr; ++q)
            std::memcpy(srec + ((size_t)kch * r0 + (size_t)(nvc + Pk::ncc(d.maxk) - 1) * nr + q) * 16 + 4 * cdw, &F.pub[d.row0 + r0 + q], 4);
        }
      }
      if (inplace) std::memcpy(P.rec.data() + (size_t)d.rec * 16, tmp.data(), rows_bytes);   // (the step pointers behind the rows stay)
      else std::memcpy(srec + (size_t)d.nrows * rs, rec + (size_t)d.nrows * rs, Pk::rec_bytes(d.nrows, d.maxk, d.nlev) - (size_t)d.nrows * rs);
      my_lds = std::max(my_lds, (((size_t)(d.nrows + d.next + 1) * sizeof(R) + 15) & ~(size_t)15) + 16);
    }
    size_t cur = lds_max.load();
    while (my_lds > cur && !lds_max.compare_exchange_weak(cur, my_lds)) {}
  });
  F.lds_max = lds_max.load();
  if (bad_dict.load() == 0 && B > 0) {
    FlowDict& D = F.dc;
    D.ent.assign((size_t)B, 0);
    size_t tot = 0;
    for (int32_t ob = 0; ob < B; ++ob) tot += dict_of[(size_t)ob].size();
    size_t drows = 0;
    for (int32_t ob = 0; ob < B; ++ob) drows += dict_of[(size_t)ob].size() / ((size_t)16 * Pk::nvc(P.blocks[ob].maxk));
    // (worth it where the rows repeat: a dictionary of at most a quarter of the rows — stencils, and Galerkin products of them)
    if (tot / 16 < ((size_t)1 << 24) && drows * 4 <= (size_t)n) {
      D.dict.reserve(tot + 16);
      for (int32_t ob = 0; ob < B; ++ob) {
        const std::vector<unsigned char>& dv = dict_of[(size_t)ob];
        const size_t vb = (size_t)16 * Pk::nvc(P.blocks[ob].maxk);
        D.ent[(size_t)ob] = (int32_t)(D.dict.size() / 16) | (int32_t)((dv.size() / vb - 1) << 24);
        D.dict.insert(D.dict.end(), dv.begin(), dv.end());
      }
      D.lds_max = F.lds_max + lds_dict_max.load();
      D.on = D.lds_max <= 64 * 1024;
    }
  }
  if (!F.dc.on) { F.dc.crec.clear(); F.dc.crec.shrink_to_fit(); }
  if (hx && bad_x.load() == 0) {
    FlowX& X = F.x;
    X.list.assign((size_t)B * 4, 0);
    size_t tot = 0;
    for (int32_t ob = 0; ob < B; ++ob) tot += xl_mb[(size_t)ob].size();
    if (tot < (size_t)INT32_MAX) {
      X.fl_mb.reserve(tot + 1); X.fl_slot.reserve(tot + 1);
      for (int32_t ob = 0; ob < B; ++ob) {
        X.list[4 * (size_t)ob] = (int32_t)X.fl_mb.size();
        X.list[4 * (size_t)ob + 1] = xl_nf[(size_t)ob];
        X.list[4 * (size_t)ob + 2] = (int32_t)xl_mb[(size_t)ob].size() - xl_nf[(size_t)ob];
        X.fl_mb.insert(X.fl_mb.end(), xl_mb[(size_t)ob].begin(), xl_mb[(size_t)ob].end());
        X.fl_slot.insert(X.fl_slot.end(), xl_slot[(size_t)ob].begin(), xl_slot[(size_t)ob].end());
      }
      if (X.fl_mb.empty()) { X.fl_mb.push_back(0); X.fl_slot.push_back(0); }
      X.row_cell_f.assign((size_t)n, -1); X.row_cell_b.assign((size_t)n, -1);
      for (int64_t q = 0; q < n; ++q) { X.row_cell_f[(size_t)P.perm[q]] = cell_f[q]; X.row_cell_b[(size_t)P.perm[q]] = cell_b[q]; }
      X.on = true;
    }
  }
  return true;
}

// ---- device side -----------------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <typename R>
struct FlowArgs {
  const Desc* blocks; const FlowDesc* fd;
  const unsigned char* srec; const uint32_t* aux;
  const int32_t* ext_col; const int32_t* fl_mb; const uint16_t* fl_slot;
  void* mbox;                        // 16 bytes per mailbox (double: {lo, epoch, hi, epoch}); 8 (float: {value, epoch})
  const R* b; R* x; R omega;
  unsigned long long* head;          // {sweeps so far << 32 | tickets drawn in the running one}: epoch = sweeps + 1
  int32_t nblocks;
  int32_t nmail;                     // mailboxes (1024 spare cells behind them)
  int32_t* err;
  long long* tim;                    // measurement hook: 4 stamps per block, then 128 step stamps per block, or null
  unsigned int spin_limit;           // polls before a wait gives up (0: the default)
  int32_t skip_pub;                  // test hook: the block of this ticket publishes nothing (a forced protocol error), -1: none
  // blocks of right-hand sides (kernel template NC > 1: a workgroup sweeps one block for a GROUP of up to NC columns — one
  // walker wave per column, one fetcher wave for all of them, every walker streaming the same record): column c's b, x and
  // mailboxes lie c * ldb, c * ldx entries and c * mail_stride bytes behind the first column's; inside the workgroup its
  // LDS x lies (c - first column of the group) * lds_stride bytes behind the first one's
  int64_t ldb = 0, ldx = 0, mail_stride = 0;
  int32_t lds_stride = 0;
  int32_t ncols = 1, ngroups = 1;    // columns of the launch, groups of <= NC columns they are swept in (grid = blocks x groups)
  int32_t xzero = 0;                 // x is zero on entry and need not be read: every block starts from zeros in LDS (the x in memory may hold anything)
  // (the relayed kernel of gs_relay.hpp only)
  uint32_t epoch = 0;                // the mailbox tag of this sweep when several launches share it (the ranks of a row-sharded level: every rank the same; 0: the launch's own count)
  const unsigned char* crec = nullptr; const unsigned char* dict = nullptr; const int32_t* dict_ent = nullptr;   // the dictionary layout (FlowDict), or null
  int32_t dict_lds = 0;              // ... and the bytes of its largest dictionary (host side only: LDS behind the columns' x)
  const int32_t* xlist = nullptr;    // the extended fetch lists of a row-sharded operator (FlowX::list; aux / fl_mb / fl_slot are then FlowX's)
  const void* rmbox = nullptr;       // mailboxes of the neighbouring rank this sweep's halo entries come from (peer-mapped; fetch-list entries with kRemoteCell set)
  int32_t late = 0;                  // the relayed kernel sums the far half of every row above the hand-over (records with the split entry layout; host side only)
  int32_t grid = 0;                  // workgroups to launch: fewer than blocks = the persistent form (host side only; 0: one per block)
};

template <typename R> struct Mail;
template <> struct Mail<double> {
  static constexpr int kBytes = 16;
  typedef u32x4 cell;
  static __device__ __forceinline__ cell load(__amdgpu_buffer_rsrc_t rs, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16); }   // sc1: past this CU's L1
  static __device__ __forceinline__ cell load_sys(__amdgpu_buffer_rsrc_t rs, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 17); }   // sc0 sc1: system scope (another device's memory)
  static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t rs, unsigned off, double v, unsigned epoch) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    cell c; c.x = (unsigned)u; c.y = epoch; c.z = (unsigned)(u >> 32); c.w = epoch;
    __builtin_amdgcn_raw_buffer_store_b128(c, rs, off, 0, 16);                                    // sc1: write-through
  }
  static __device__ __forceinline__ bool valid(const cell& c, unsigned epoch) { return c.y == epoch && c.w == epoch; }
  static __device__ __forceinline__ double value(const cell& c) { return __longlong_as_double((long long)(((unsigned long long)c.z << 32) | c.x)); }
};
template <> struct Mail<float> {
  static constexpr int kBytes = 8;
  typedef u32x2 cell;
  static __device__ __forceinline__ cell load(__amdgpu_buffer_rsrc_t rs, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 16); }
  static __device__ __forceinline__ cell load_sys(__amdgpu_buffer_rsrc_t rs, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 17); }
  static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t rs, unsigned off, float v, unsigned epoch) {
    cell c; c.x = __float_as_uint(v); c.y = epoch;
    __builtin_amdgcn_raw_buffer_store_b64(c, rs, off, 0, 16);
  }
  static __device__ __forceinline__ bool valid(const cell& c, unsigned epoch) { return c.y == epoch; }
  static __device__ __forceinline__ float value(const cell& c) { return __uint_as_float(c.x); }
};

typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int NDW> struct Dw;
template <> struct Dw<1> { typedef unsigned int type; };
template <> struct Dw<2> { typedef u32x2 type; };
template <> struct Dw<3> { typedef u32x3 type; };
template <> struct Dw<4> { typedef u32x4 type; };
__device__ __forceinline__ u32x4 widen(unsigned int v) { u32x4 r = {v, 0u, 0u, 0u}; return r; }
__device__ __forceinline__ u32x4 widen(u32x2 v) { u32x4 r = {v.x, v.y, 0u, 0u}; return r; }
__device__ __forceinline__ u32x4 widen(u32x3 v) { u32x4 r = {v.x, v.y, v.z, 0u}; return r; }
__device__ __forceinline__ u32x4 widen(u32x4 v) { return v; }
// The walker's memory operations are written out: the compiler counts the loads it schedules itself, but in this loop its
// counting collapses (any load issued before a loop it cannot analyse, any branch with a store inside, makes it wait for
// nearly everything in flight at every step), and the pipeline below lives on exact counts.  A load's destination is a
// compiler-allocated register ("=v"); nothing reads it before the flow_wait statement that names it ("+v").
__device__ __forceinline__ i32x4 make_rsrc(const void* base) {
  const unsigned long long u = (unsigned long long)base;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)u);
  r.y = __builtin_amdgcn_readfirstlane((int)((u >> 32) & 0xffffu));
  r.z = 0x7ffffff0; r.w = 0x00020000;
  return r;
}
template <int NDW>
__device__ __forceinline__ void asm_load(typename Dw<NDW>::type& r, unsigned voff, i32x4 rs, unsigned soff) {
  if constexpr (NDW == 4) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r) : "v"(voff), "s"(rs), "s"(soff) : "memory");
  else if constexpr (NDW == 3) asm volatile("buffer_load_dwordx3 %0, %1, %2, %3 offen" : "=v"(r) : "v"(voff), "s"(rs), "s"(soff) : "memory");
  else if constexpr (NDW == 2) asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(r) : "v"(voff), "s"(rs), "s"(soff) : "memory");
  else asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(r) : "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ void asm_store_sc1(u32x4 v, unsigned voff, i32x4 rs) {   // write-through; (s_nop: the data registers are read after issue)
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen sc1\n\ts_nop 1" :: "v"(v), "v"(voff), "s"(rs) : "memory");
}
__device__ __forceinline__ void asm_store_sc1(u32x2 v, unsigned voff, i32x4 rs) {
  asm volatile("buffer_store_dwordx2 %0, %1, %2, 0 offen sc1" :: "v"(v), "v"(voff), "s"(rs) : "memory");
}
__device__ __forceinline__ void asm_store(u32x2 v, unsigned voff, i32x4 rs) {
  asm volatile("buffer_store_dwordx2 %0, %1, %2, 0 offen" :: "v"(v), "v"(voff), "s"(rs) : "memory");
}
__device__ __forceinline__ void asm_store(unsigned int v, unsigned voff, i32x4 rs) {
  asm volatile("buffer_store_dword %0, %1, %2, 0 offen" :: "v"(v), "v"(voff), "s"(rs) : "memory");
}

template <typename R, int MAXK>
struct FlowOps {
  static constexpr int VPC = 16 / (int)sizeof(R), NVC = (MAXK + 2 + VPC - 1) / VPC, NCC = (MAXK + 7) / 8, KCH = (NVC + NCC) | 1;
  // dwords of the last value / column chunk that carry data (a load fetches exactly those)
  static constexpr int VTAIL = ((MAXK + 2) * (int)sizeof(R) / 4 - 1) % 4 + 1, CTAIL = ((MAXK + 1) / 2 - 1) % 4 + 1;
  static_assert(CTAIL < 4, "the publish word rides in the spare dword behind the row's last columns");
  static constexpr int NLOAD = NVC + NCC + 1;   // loads per step: the chunks and b
  u32x4 vf[NVC > 1 ? NVC - 1 : 1];              // full value chunks
  typename Dw<VTAIL>::type vt;                  // the last one: ... diagonal, reciprocal
  u32x4 cf[NCC > 1 ? NCC - 1 : 1];              // full column chunks
  typename Dw<CTAIL + 1>::type ct;              // the last one: columns, publish word
  typename Dw<(int)sizeof(R) / 4>::type bb;
  int p; int need;
  __device__ __forceinline__ u32x4 vchunk(int c) const { return c < NVC - 1 ? vf[c] : widen(vt); }
  __device__ __forceinline__ u32x4 cchunk(int c) const { return c < NCC - 1 ? cf[c] : widen(ct); }
  __device__ __forceinline__ int32_t pub() const { const u32x4 c = widen(ct); return (int32_t)(CTAIL == 1 ? c.y : CTAIL == 2 ? c.z : c.w); }
};
// every load of o has landed once at most N younger memory operations are in flight
template <int N, typename O>
__device__ __forceinline__ void flow_wait(O& o) {
  asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory");
#pragma unroll
  for (int c = 0; c < O::NVC - 1; ++c) asm volatile("" : "+v"(o.vf[c]));
  asm volatile("" : "+v"(o.vt));
#pragma unroll
  for (int c = 0; c < O::NCC - 1; ++c) asm volatile("" : "+v"(o.cf[c]));
  asm volatile("" : "+v"(o.ct));
  asm volatile("" : "+v"(o.bb));
}
template <typename R, typename O> __device__ __forceinline__ R chunk_value(const O& o, int k) {   // value k of the row (k: a constant after unrolling)
  if constexpr (sizeof(R) == 8) {
    const u32x4 c = o.vchunk(k >> 1);
    return (k & 1) ? __hiloint2double((int)c.w, (int)c.z) : __hiloint2double((int)c.y, (int)c.x);
  } else {
    const u32x4 c = o.vchunk(k >> 2);
    const unsigned w = (k & 3) == 0 ? c.x : (k & 3) == 1 ? c.y : (k & 3) == 2 ? c.z : c.w;
    return __uint_as_float(w);
  }
}
// a step's operands on the DICTIONARY layout (FlowDict): column chunks and b only — the values come out of the block's
// dictionary in LDS, by the index that rides in the publish word
template <typename R, int MAXK>
struct FlowOpsD {
  static constexpr int VPC = 16 / (int)sizeof(R), NVC = (MAXK + 2 + VPC - 1) / VPC, NCC = (MAXK + 7) / 8, KCH = NCC;
  static constexpr int CTAIL = ((MAXK + 1) / 2 - 1) % 4 + 1;
  static constexpr int NLOAD = NCC + 1;
  u32x4 cf[NCC > 1 ? NCC - 1 : 1];
  typename Dw<CTAIL + 1>::type ct;
  typename Dw<(int)sizeof(R) / 4>::type bb;
  int p; int need;
  __device__ __forceinline__ u32x4 cchunk(int c) const { return c < NCC - 1 ? cf[c] : widen(ct); }
  __device__ __forceinline__ int32_t pub() const { const u32x4 c = widen(ct); return (int32_t)(CTAIL == 1 ? c.y : CTAIL == 2 ? c.z : c.w); }
};
template <int N, typename R, int MAXK>
__device__ __forceinline__ void flow_wait(FlowOpsD<R, MAXK>& o) {
  asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory");
#pragma unroll
  for (int c = 0; c < FlowOpsD<R, MAXK>::NCC - 1; ++c) asm volatile("" : "+v"(o.cf[c]));
  asm volatile("" : "+v"(o.ct));
  asm volatile("" : "+v"(o.bb));
}
template <typename R> __device__ __forceinline__ R value_of(const u32x4* vv, int k) {   // value k of a row out of its value chunks
  if constexpr (sizeof(R) == 8) {
    const u32x4 c = vv[k >> 1];
    return (k & 1) ? __hiloint2double((int)c.w, (int)c.z) : __hiloint2double((int)c.y, (int)c.x);
  } else {
    const u32x4 c = vv[k >> 2];
    const unsigned w = (k & 3) == 0 ? c.x : (k & 3) == 1 ? c.y : (k & 3) == 2 ? c.z : c.w;
    return __uint_as_float(w);
  }
}

template <typename R> __device__ __forceinline__ R load_real(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff);
template <> __device__ __forceinline__ double load_real<double>(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  const auto t = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
  return __hiloint2double((int)t[1], (int)t[0]);
}
template <> __device__ __forceinline__ float load_real<float>(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
__device__ __forceinline__ void store_real(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, double v) {
  u32x2 t; t.x = (unsigned)__double2loint(v); t.y = (unsigned)__double2hiint(v);
  __builtin_amdgcn_raw_buffer_store_b64(t, rs, voff, soff, 0);
}
__device__ __forceinline__ void store_real(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, voff, soff, 0);
}

// a pointer every lane holds alike, moved into scalar registers (a buffer descriptor the compiler cannot prove uniform is
// wrapped in a serialising "waterfall" loop around every access)
template <typename T> __device__ __forceinline__ T* uniform_ptr(T* p) {
  const unsigned long long u = (unsigned long long)p;
  return (T*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)u));
}
// The fetcher's arrival at the workgroup's barrier, as an instruction of its own: written as __syncthreads() the compiler
// sinks it into ONE barrier block shared with the walkers' — correct, but every path through the fetcher's code then starts
// in a block the walkers' hand-issued loads flow through, and the whole-kernel audit (tools/flow_asm_linear.py) can no longer
// tell the two apart.  (LDS writes done, then the barrier; the fetcher has no stores to memory before it.)
__device__ __forceinline__ void fetcher_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\t; (the fetcher's arrival)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// (blocks of right-hand sides: the fetcher's column arrays share the 128 registers of the short-row kernels — one set less)
template <int MAXK, int NC = 1> struct FlowDepth {
  static constexpr int value = MAXK <= 6 ? (NC > 1 && BW_FLOW_DEPTH > 3 ? 3 : BW_FLOW_DEPTH) : MAXK <= 12 ? (BW_FLOW_DEPTH < 4 ? BW_FLOW_DEPTH : 4) : 3;
};
static_assert(BW_FLOW_DEPTH >= 2 && BW_FLOW_DEPTH <= 6, "the first round of the pipeline is written out for up to six sets");

template <int MAXK> struct FlowWaves { static constexpr int value = MAXK <= 6 ? BW_FLOW_WAVES : 2; };   // waves per SIMD the kernel is compiled for
// (the dictionary layout: a set in flight is a column chunk or three and b)
#ifndef BW_FLOW_DICT_DEPTH
#define BW_FLOW_DICT_DEPTH 3
#endif
template <int MAXK> struct FlowWavesD { static constexpr int value = MAXK <= 6 ? 5 : 3; };

// One workgroup = one block of the partition x one group of up to NC right-hand-side columns: waves 0 .. NC - 1 WALK the block,
// one column each (own LDS x, own mailboxes, all of them streaming the same record: the first wave's loads bring it into the
// CU's cache), wave NC FETCHES for all of them.  NC = 1 is the single-column sweep.
template <typename R, bool SOR, bool BWD, int MAXK, int NC = 1, bool DICT = false>
__global__ __launch_bounds__(64 * (NC + 1), DICT ? FlowWavesD<MAXK>::value : FlowWaves<MAXK>::value) void gs_bw_flow_kernel(FlowArgs<R> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_all[];
  static_assert(NC >= 1 && NC + 1 <= 4 * FlowWaves<MAXK>::value, "the workgroup's waves must fit the registers the kernel is compiled for");
  typedef typename std::conditional<DICT, FlowOpsD<R, MAXK>, FlowOps<R, MAXK>>::type O;
  typedef Mail<R> M;
  constexpr int D = DICT ? BW_FLOW_DICT_DEPTH : FlowDepth<MAXK, NC>::value;
  constexpr int NVC = FlowOps<R, MAXK>::NVC;
  constexpr int RB = (int)sizeof(R);
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = (int)(threadIdx.x & 63u);
  const bool fetcher = wv == NC;
  const long long t_start = a.tim ? wall_clock64() : 0;
  // (the ticket travels through the first word of the dynamic LDS — no static LDS: the base stays 16-byte aligned)
  unsigned long long* s_ticket = (unsigned long long*)lds_all;
  // (a launch starts with every resident workgroup at this line at once; one word takes ~88 atomics per microsecond, and the
  // first blocks' loads queue behind the burst — so the first workgroups stagger their draw, 12 ns apart: far less than the
  // time before their blocks can run.  A pure delay: tickets are still drawn by whoever comes)
  if (blockIdx.x < 2048u) for (unsigned i = 0; i < (blockIdx.x >> 5); ++i) __builtin_amdgcn_s_sleep(14);
  if (threadIdx.x == 0) *s_ticket = atomicAdd(a.head, 1ull);
  __syncthreads();
  // (made wave-uniform by hand: everything derived from it — descriptors, step words, buffer offsets — then lives in scalar
  // registers; left to the compiler, a value read from LDS counts as divergent and every buffer access becomes a waterfall loop)
  const unsigned long long tv = *s_ticket;
  const unsigned long long ticket = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(tv >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)tv);
  __syncthreads();   // (the word is x[0] of the first column from here on)
  // tickets run over (block, column group), the groups of a block side by side: a unit only ever waits for units of its own
  // group with smaller tickets.  The word holds {sweeps so far, units drawn}; whoever draws the launch's last ticket starts
  // the next sweep's count (every other draw of this launch has happened by then; launches of different shapes may follow
  // each other on the same word)
  const unsigned int units = (unsigned int)a.nblocks * (unsigned int)a.ngroups;
  const unsigned int ut = (unsigned int)ticket, sweeps = (unsigned int)(ticket >> 32);
  const unsigned int epoch = sweeps + 1u;
  if (threadIdx.x == 0 && ut == units - 1u)
    __hip_atomic_store(a.head, (unsigned long long)(sweeps + 2u >= 0x80000000u ? 0u : sweeps + 1u) << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int tk = (int)(ut / (unsigned)a.ngroups);
  const int c0 = (int)(ut % (unsigned)a.ngroups) * NC;
  const int nch = a.ncols - c0 < NC ? a.ncols - c0 : NC;   // columns of this group
  const int ob = __builtin_amdgcn_readfirstlane(BWD ? a.nblocks - 1 - tk : tk);
  const Desc d = a.blocks[ob];
  const FlowDesc f = a.fd[ob];
  const int ns = d.nlev;
  const int nxb = (int)(((size_t)(d.nrows + d.next + 1) * sizeof(R) + 15) & ~(size_t)15);
  const unsigned spin_limit = a.spin_limit ? a.spin_limit : (1u << 22);
  // column c of the group: b, x, mailboxes and LDS x
  auto col_x = [&](int c) { return a.x + (int64_t)(c0 + c) * a.ldx; };
  auto col_mail = [&](int c) { return (void*)((unsigned char*)a.mbox + (int64_t)(c0 + c) * a.mail_stride); };
  auto col_lds = [&](int c) { return lds_all + (size_t)c * (size_t)a.lds_stride; };
  if (fetcher) {
    // ---- load phase, the fetcher's half (the walkers bring their own rows, below): for every column of the group the
    // external columns of the far side as they stand (the columns the sweep has not reached — behind the block going
    // forward, before it going backward — and halo columns; the near side arrives through the mailboxes).  The positions
    // are the same for every column: read once, then the gathers of all columns of a batch in flight at once ----
    constexpr int EU = 4;
    const R* xc[NC]; R* xlc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { xc[c] = uniform_ptr(col_x(c < nch ? c : 0)); xlc[c] = (R*)col_lds(c); }
    const int nfar = BWD ? d.npre + (d.next - d.npre - f.npost) : d.next - d.npre;
    if (a.xzero) {   // (x = 0 everywhere: the far side is zeros, nothing to gather)
      for (int e0 = 0; e0 < d.next; e0 += 64) {
        const int i = e0 + lane;
#pragma unroll
        for (int c = 0; c < NC; ++c) if (i < d.next && c < nch) xlc[c][d.nrows + i] = (R)0;
      }
    } else
    for (int e0 = 0; e0 < nfar; e0 += 64 * EU) {
      int32_t ec[EU]; int es[EU];
#pragma unroll
      for (int k = 0; k < EU; ++k) {
        const int i = e0 + lane + 64 * k;
        es[k] = BWD ? (i < d.npre ? i : i + f.npost) : d.npre + i;   // slot in the external list
        ec[k] = a.ext_col[i < nfar ? d.ext0 + es[k] : 0];
      }
#pragma unroll
      for (int k = 0; k < EU; ++k) pin(ec[k]);
      R xe[NC][EU];
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int k = 0; k < EU; ++k) xe[c][k] = xc[c][ec[k]];
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int k = 0; k < EU; ++k) pin(xe[c][k]);
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int k = 0; k < EU; ++k) { const int i = e0 + lane + 64 * k; if (i < nfar && c < nch) xlc[c][d.nrows + es[k]] = xe[c][k]; }
    }
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < NC; ++c) if (c < nch) { xlc[c][d.nrows + d.next] = (R)0; *(unsigned int*)(col_lds(c) + nxb) = 0u; }
    }
    // (the fetcher's path is its own from here to its end — its own arrival at the block's barrier: nothing of the walkers'
    // hand-counted loads is in flight on any path through it, which is what tools/flow_asm_linear.py can then see)
    fetcher_barrier();
    {
      // ---- the fetcher: near-side values out of their mailboxes, in the order the walk needs them, for every column of the
      // group.  A window of U x 64 list entries is polled (per column only the groups up to the one behind its first missing
      // entry); a column's progress word counts the LEADING entries that have arrived, so its walker goes on as soon as
      // what its next step reads is there ----
      constexpr int U = NC <= 2 ? 3 : 2;
      const int nf = BWD ? f.npost : d.npre;
      const int f0 = d.ext0 + (BWD ? d.npre : 0);
      __amdgpu_buffer_rsrc_t rs_mail[NC];
      R* xlc[NC]; unsigned int* prog[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        rs_mail[c] = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(col_mail(c < nch ? c : 0)), 0, 0x7ffffff0, 0x00020000);
        xlc[c] = (R*)col_lds(c); prog[c] = (unsigned int*)(col_lds(c) + nxb);
      }
      unsigned total_spins = 0;
      bool fail = false;
      for (int w0 = 0; w0 < nf && !fail; w0 += 64 * U) {
        int32_t mb[U]; int slot[U]; bool ok[NC][U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e = w0 + 64 * u + lane;
          mb[u] = a.fl_mb[e < nf ? f0 + e : f0];
          slot[u] = a.fl_slot[e < nf ? f0 + e : f0];
#pragma unroll
          for (int c = 0; c < NC; ++c) ok[c][u] = !(e < nf) || !(c < nch);
        }
        int first[NC];   // leading entries of the window that have arrived, per column
#pragma unroll
        for (int c = 0; c < NC; ++c) first[c] = c < nch ? 0 : 64 * U;
        for (;;) {
          typename M::cell cl[NC][U];
#pragma unroll
          for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int u = 0; u < U; ++u) if (!ok[c][u] && 64 * u <= first[c] + 64) cl[c][u] = M::load(rs_mail[c], (unsigned)mb[u] * (unsigned)M::kBytes);
          bool moved = false, done = true;
#pragma unroll
          for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int u = 0; u < U; ++u)
              if (!ok[c][u] && 64 * u <= first[c] + 64 && M::valid(cl[c][u], epoch)) { xlc[c][slot[u]] = M::value(cl[c][u]); ok[c][u] = true; }
            int nfirst = 64 * U;
#pragma unroll
            for (int u = U - 1; u >= 0; --u) {
              const unsigned long long m = __builtin_amdgcn_ballot_w64(!ok[c][u]);
              if (m) nfirst = 64 * u + (int)__builtin_ctzll(m);
            }
            // (LDS operations of one wave execute in program order: the values are in place before the progress word moves)
            if (nfirst != first[c] && lane == 0) __hip_atomic_store(prog[c], (unsigned)(w0 + nfirst < nf ? w0 + nfirst : nf), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            moved |= nfirst != first[c];
            first[c] = nfirst;
            done &= nfirst == 64 * U;
          }
          if (done) break;
          if (!moved) __builtin_amdgcn_s_sleep(4);
          if (++total_spins > spin_limit) { fail = true; break; }
        }
      }
      if (fail && lane == 0) {
        *a.err = 1;
#pragma unroll
        for (int c = 0; c < NC; ++c) if (c < nch) __hip_atomic_store(prog[c], 0x7fffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      return;
    }
  }
  // (walkers of columns the group does not have leave behind the block's barrier)
  // (... by a path of their own, like the fetcher: every path below this line is a walker's)
  if (wv >= nch) { fetcher_barrier(); return; }
  const int wc = wv;
  unsigned char* lds = col_lds(wc);
  R* xl = (R*)lds;
  unsigned int* progress = (unsigned int*)(lds + nxb);
  R* x = col_x(wc);
  const R* bcol = a.b + (int64_t)(c0 + wc) * a.ldb;
  void* mcol = col_mail(wc);
  // the walker's pipeline starts ahead of the block's load phase: step words, then the operands of the first D steps
  const uint32_t* ax = a.aux + f.aux + (BWD ? ns + 1 : 0);
  // (step words by scalar loads, one issue ahead: a vector-loaded register read inside the main loop makes the compiler
  // drain every load in flight there — its counting loses track of a load issued before a loop it cannot analyse)
  typedef const uint32_t __attribute__((address_space(4))) cu32;   // (constant address space: read-only for the kernel's lifetime, so a uniform index is a scalar load)
  const cu32* axc = (const cu32*)(unsigned long long)uniform_ptr(ax);
  unsigned sw_next = axc[0];
  const i32x4 rs_rec = DICT ? make_rsrc(a.crec + (size_t)(uint32_t)d.row0 * (size_t)(16 * O::NCC)) : make_rsrc(a.srec + (size_t)(uint32_t)d.rec * 16);
  const i32x4 rs_b = make_rsrc(bcol + d.row0);
  const i32x4 rs_x = make_rsrc(x + d.row0);
  const i32x4 rs_mst = make_rsrc(mcol);
  // operands of walking step kk into o (steps behind the last one: a harmless re-read of the block's first chunk);
  // O::NLOAD loads, always
  auto issue = [&](O& o, int kk) {
    const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)sw_next);   // the word of step min(kk, ns): the steps are issued in order
    sw_next = axc[kk + 1 < ns ? kk + 1 : ns];
    const int r0 = (int)(w & ((1u << kStepRowBits) - 1)), nr = (int)((w >> kStepRowBits) & ((1u << kStepCntBits) - 1));
    o.need = (int)(w >> (kStepRowBits + kStepCntBits));
    // (idle lanes repeat the step's first row: the same reads, the same value to the same places — no branch in a step)
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
  // (the dictionary layout: the block's distinct value rows, behind the columns' LDS; every walker brings a share)
  const u32x4* dict_lds = (const u32x4*)(lds_all + (size_t)NC * (size_t)a.lds_stride);
  {
    if constexpr (DICT) {
      const int32_t de = a.dict_ent[ob];
      const int nchunk = (int)(((unsigned)de >> 24) + 1u) * NVC;
      const u32x4* dg = (const u32x4*)(a.dict + (size_t)(de & 0xffffff) * 16);
      u32x4* dl = (u32x4*)(lds_all + (size_t)NC * (size_t)a.lds_stride);
      for (int i = wv * 64 + lane; i < nchunk; i += 64 * nch) dl[i] = dg[i];
    }
    // the walker's half of the load phase: its column's own rows into LDS (one batch of loads for blocks of up to 512 rows),
    // finished — the compiler waits for them before the LDS writes — before the hand-counted pipeline starts
    constexpr int XW = 8;
    if (a.xzero) {
      for (int p = lane; p < d.nrows; p += 64) xl[p] = (R)0;
    } else
    for (int p0 = 0; p0 < d.nrows; p0 += 64 * XW) {
      R xv[XW];
#pragma unroll
      for (int k = 0; k < XW; ++k) { const int p = p0 + lane + 64 * k; xv[k] = x[d.row0 + (p < d.nrows ? p : 0)]; }
#pragma unroll
      for (int k = 0; k < XW; ++k) pin(xv[k]);
#pragma unroll
      for (int k = 0; k < XW; ++k) { const int p = p0 + lane + 64 * k; if (p < d.nrows) xl[p] = xv[k]; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (nothing of the compiler's in flight when the counted loads start)
#pragma unroll
    for (int j = 0; j < D; ++j) issue(ops[j], j);
  }
  __syncthreads();
  // ---- the walker ----
  __builtin_amdgcn_s_setprio(2);
  const long long t_loaded = a.tim ? wall_clock64() : 0;
  const int32_t pubdir = BWD ? kPubBwd : kPubFwd;
  const bool mute = a.skip_pub == tk;
  const unsigned spare = (unsigned)a.nmail + (unsigned)(ob & 1023);   // (behind the mailboxes: 1024 cells nobody reads)
  unsigned seen = 0;
  bool gave_up = false;
  auto step = [&](const O& o, int kk) {
    // the near-side values this step reads are in LDS
    if ((unsigned)o.need > seen) {
      unsigned spins = 0;
      for (;;) {
        seen = __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (seen >= (unsigned)o.need) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 4u * spin_limit) { gave_up = true; break; }
      }
    }
    asm volatile("" ::: "memory");   // the gathers stay below the progress word
    R xv[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
      const u32x4 cw = o.cchunk(k >> 3);
      const unsigned w = ((k >> 1) & 3) == 0 ? cw.x : ((k >> 1) & 3) == 1 ? cw.y : ((k >> 1) & 3) == 2 ? cw.z : cw.w;
      const unsigned off = (k & 1) ? (w >> 16) : (w & 0xffffu);
      xv[k] = *(const R*)((const char*)xl + off);
    }
    const R xo = xl[o.p];
    u32x4 vv[NVC];
    if constexpr (DICT) {
      const u32x4* dr = dict_lds + (((unsigned)o.pub() >> kDictIdxShift) & 0xffu) * (unsigned)NVC;
#pragma unroll
      for (int c = 0; c < NVC; ++c) vv[c] = dr[c];
    }
    auto val = [&](int k) -> R { if constexpr (DICT) return value_of<R>(vv, k); else return chunk_value<R>(o, k); };
    R acc = (R)0;
#pragma unroll
    for (int k = 0; k < MAXK; ++k) acc += val(k) * xv[k];
    const R dg = val(MAXK), rc = val(MAXK + 1);
    R bbv;
    if constexpr (sizeof(R) == 8) bbv = __hiloint2double((int)o.bb.y, (int)o.bb.x); else bbv = __uint_as_float(o.bb);
    const R nn = bbv - acc;
    R q;
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
    q = dg != (R)0 ? q : xo;   // a zero diagonal: the row keeps its value (smoother.jl:87) — and publishes it: its readers wait for the tag
    xl[o.p] = q;
    // a row another block reads: into its mailbox at once, tagged with the sweep's epoch; every other lane into the block's
    // spare cell (one more line per step instead of a branch around the store)
    const int32_t pw = o.pub();
    const unsigned cell = ((pw & pubdir) && !mute) ? (unsigned)f.pad0 + (((unsigned)pw >> (BWD ? 11 : 0)) & 0x7ffu) : spare;
    if constexpr (sizeof(R) == 8) {
      const unsigned qlo = (unsigned)__double2loint(q), qhi = (unsigned)__double2hiint(q);
      u32x4 cv = {qlo, epoch, qhi, epoch};
      asm_store_sc1(cv, cell * 16u, rs_mst);
      u32x2 xv2 = {qlo, qhi};
      asm_store(xv2, (unsigned)o.p * 8u, rs_x);
    } else {
      u32x2 cv = {__float_as_uint(q), epoch};
      asm_store_sc1(cv, cell * 8u, rs_mst);
      asm_store(__float_as_uint(q), (unsigned)o.p * 4u, rs_x);
    }
#ifdef BW_FLOW_STEP_STAMPS   // (measurement builds only: a conditional store in the step makes the compiler's load counting conservative)
    if (a.tim && lane == 0) a.tim[4 * (int64_t)a.nblocks + 128 * (int64_t)ob + kk] = wall_clock64();
#endif
  };
  // The pipeline: D register sets; a step waits for its own set (vmcnt counted exactly: between a set's loads and its use lie
  // the loads and the two stores of the D - 1 steps in between — fewer at the start, where the sets were loaded back to back),
  // computes, stores, and re-loads the set for the step D ahead.
  constexpr int L = O::NLOAD, S = 2;
  int k = 0;
  if (ns >= D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {   // the first round: set j was followed by D - 1 - j more sets and j whole steps
      if (j == 0) flow_wait<(D - 1) * L>(ops[0]);
      else if (j == 1) flow_wait<(D - 2) * L + 1 * (L + S)>(ops[1 < D ? 1 : 0]);
      else if (j == 2) flow_wait<(D > 2 ? D - 3 : 0) * L + 2 * (L + S)>(ops[2 < D ? 2 : 0]);
      else if (j == 3) flow_wait<(D > 3 ? D - 4 : 0) * L + 3 * (L + S)>(ops[3 < D ? 3 : 0]);
      else if (j == 4) flow_wait<(D > 4 ? D - 5 : 0) * L + 4 * (L + S)>(ops[4 < D ? 4 : 0]);
      else flow_wait<(D > 5 ? D - 6 : 0) * L + 5 * (L + S)>(ops[5 < D ? 5 : 0]);
      step(ops[j], j);
      issue(ops[j], j + D);
    }
    for (k = D; k + D <= ns; k += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        flow_wait<(D - 1) * (L + S)>(ops[j]);
        step(ops[j], k + j);
        issue(ops[j], k + j + D);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < D - 1; ++j)
    if (k + j < ns) { flow_wait<0>(ops[j]); step(ops[j], k + j); }
  if (gave_up && lane == 0) *a.err = 2;
  if (a.tim && lane == 0 && wv == 0 && c0 == 0) {
    long long* tt = a.tim + 4 * (int64_t)ob;
    tt[0] = t_start; tt[1] = t_loaded; tt[2] = wall_clock64(); tt[3] = 0;
  }
}

// columns one workgroup sweeps (walker waves beside the one fetcher): as many as leave three (rows of <= 6 entries: 128
// registers per lane, 16 waves per CU) / two (longer rows: 256 registers, 8 waves) workgroups resident per CU
template <int MAXK> struct FlowMaxNc { static constexpr int value = MAXK <= 6 ? 4 : 3; };
constexpr size_t kFlowLdsCap = 152 * 1024;

template <typename R, bool SOR, bool BWD, int MAXK, int NC, bool DICT = false>
inline hipError_t sweep_flow_launch(const FlowArgs<R>& a, size_t lds, hipStream_t st) {
  if constexpr (NC > FlowMaxNc<MAXK>::value) return hipErrorInvalidValue;
  else {
    auto* fn = gs_bw_flow_kernel<R, SOR, BWD, MAXK, NC, DICT>;
    if (lds > 64 * 1024) {   // (beyond the default limit: asked for once per kernel)
      static hipError_t once = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFlowLdsCap);
      if (once != hipSuccess) return once;
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)a.nblocks * (unsigned)a.ngroups), dim3(64 * (NC + 1)), lds, st, a);
    return hipGetLastError();
  }
}
template <typename R, int MAXK, int NC>
inline hipError_t sweep_flow_k(const FlowArgs<R>& a, size_t lds, bool sor, bool backward, hipStream_t st) {
  if (a.crec) {   // (the dictionary layout, where the schedule carries one)
    {
      if (sor) return backward ? sweep_flow_launch<R, true, true, MAXK, NC, true>(a, lds, st) : sweep_flow_launch<R, true, false, MAXK, NC, true>(a, lds, st);
      return backward ? sweep_flow_launch<R, false, true, MAXK, NC, true>(a, lds, st) : sweep_flow_launch<R, false, false, MAXK, NC, true>(a, lds, st);
    }
  }
  if (sor) return backward ? sweep_flow_launch<R, true, true, MAXK, NC>(a, lds, st) : sweep_flow_launch<R, true, false, MAXK, NC>(a, lds, st);
  return backward ? sweep_flow_launch<R, false, true, MAXK, NC>(a, lds, st) : sweep_flow_launch<R, false, false, MAXK, NC>(a, lds, st);
}
// ncols columns (a.ldb / a.ldx / a.mail_stride apart) in ONE launch: groups of equal size, at most nc_max columns each
template <typename R, int MAXK>
inline hipError_t sweep_flow_cols(FlowArgs<R> a, size_t lds_max, bool sor, bool backward, int ncols, int nc_max, hipStream_t st) {
  const size_t stride = (lds_max + 15) & ~(size_t)15;
  if (stride > 64 * 1024 || ncols < 1) return hipErrorInvalidValue;
  if ((unsigned long long)a.nblocks * (unsigned long long)ncols > 0x7fffffffull) return hipErrorInvalidValue;
  int cap = FlowMaxNc<MAXK>::value;
  if (nc_max > 0 && nc_max < cap) cap = nc_max;
  const size_t dl = a.crec ? (size_t)a.dict_lds : 0;   // (the block's dictionary behind the columns)
  while (cap > 1 && (size_t)cap * stride + dl > kFlowLdsCap) --cap;
  const int groups = (ncols + cap - 1) / cap, nc = (ncols + groups - 1) / groups;
  a.lds_stride = (int32_t)stride; a.ncols = ncols; a.ngroups = groups;
  switch (nc) {
    case 4: return sweep_flow_k<R, MAXK, 4>(a, stride * 4 + dl, sor, backward, st);
    case 3: return sweep_flow_k<R, MAXK, 3>(a, stride * 3 + dl, sor, backward, st);
    case 2: return sweep_flow_k<R, MAXK, 2>(a, stride * 2 + dl, sor, backward, st);
    default: return sweep_flow_k<R, MAXK, 1>(a, stride + dl, sor, backward, st);
  }
}
template <typename R>
inline hipError_t sweep_flow(const FlowArgs<R>& a, int maxk, size_t lds_max, bool sor, bool backward, hipStream_t st, int ncols = 1, int nc_max = 0) {
  switch (maxk) {
    case 6: return sweep_flow_cols<R, 6>(a, lds_max, sor, backward, ncols, nc_max, st);
    case 12: return sweep_flow_cols<R, 12>(a, lds_max, sor, backward, ncols, nc_max, st);
#if BW_PLAN_MAXK >= 18
    case 18: return sweep_flow_cols<R, 18>(a, lds_max, sor, backward, ncols, nc_max, st);
#endif
  }
  return hipErrorInvalidValue;
}

}  // namespace bw
}  // namespace amgh
