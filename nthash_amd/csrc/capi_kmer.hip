// capi_kmer.hip -- nthip_kmer_hash: validation and path selection
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
#include "capi_internal.hpp"
#include "util_kernels.hpp"

using namespace ntamd;
using namespace ntamd::host;

extern "C" int nthip_kmer_hash(nthip_ctx* c, const nthip_reads* rd_in, uint16_t k16, uint8_t m8,
                               const nthip_out* out, uint64_t* total_out, uint32_t flags)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  NTCHK(check_reads(rd_in));
  nthip_reads eff = *rd_in; // what the paths below see: offsets of equal-length, back-to-back reads become a fixed length
  const nthip_reads* rd = &eff;
  if (!out || !out->hashes) return fail(NTHIP_ERR_ARG, "out->hashes is NULL");
  const uint32_t k = k16, m = m8;
  if (k == 0) return fail(NTHIP_ERR_ARG, "k must be greater than 0"); // src/kmer.cpp:212-214
  if (k < 3) return fail(NTHIP_ERR_UNSUPPORTED, "k < 3 is undefined in the reference (src/kmer.cpp:47)");
  if (m == 0) return fail(NTHIP_ERR_UNSUPPORTED, "num_hashes must be >= 1");
  HIPCHK(hipSetDevice(c->device));
  uint64_t total = 0;
  if (total_out) *total_out = 0;
  if (rd->n_reads == 0) return NTHIP_OK;

  if (flags & NTHIP_PACKED_INPUT) { // reads->seqs is what nthip_pack_reads made (capi_packed.hip)
    Staged pst;
    NTCHK(stage_outputs(c, out, flags, rd->n_reads, m, pst));
    const int rc = run_kmer_packed(c, rd, k, m, out, pst, flags, &total);
    if (total_out) *total_out = total;
    NTCHK(rc);
    NTCHK(unstage_outputs(c, out, flags, rd->n_reads, m, total, pst));
    HIPCHK(hipStreamSynchronize(c->stream));
    return NTHIP_OK;
  }

  uint64_t total_bytes = 0;
  NTCHK(reads_total_bytes(c, rd, flags, &total_bytes));
  Staged st;
  NTCHK(stage_inputs(c, rd, flags, total_bytes, st));
  NTCHK(stage_outputs(c, out, flags, rd->n_reads, m, st));

  // NTHIP_OUT_READ_SLOTS on fixed-length reads (round 3): slot r = r * (len - k + 1).  The dense pass runs as if the batch
  // were clean, marking the 16-byte vectors that hold a non-base; the reads those touch are redone afterwards
  // (kmer_fixed_slots_finish).  A batch with an N in one read of a thousand then costs what a clean one does.
  const bool fslots = (flags & NTHIP_OUT_READ_SLOTS) && !st.offsets;
  if (fslots) {
    const uint32_t flen = rd->fixed_len, fstride = rd->stride ? rd->stride : flen;
    if (flags & (NTHIP_ASYNC | NTHIP_FORCE_GENERAL | NTHIP_FORCE_ROWS | NTHIP_PACKED_INPUT))
      return fail(NTHIP_ERR_UNSUPPORTED, "NTHIP_OUT_READ_SLOTS: a plain synchronous call on ASCII reads");
    if (!st.counts || st.fwd || st.rev) return fail(NTHIP_ERR_ARG, "NTHIP_OUT_READ_SLOTS needs out->counts and has no strand outputs");
    if (fstride != flen || !kmer_fixed_slots_len_ok(flen) || kmer_runs_chunked_compiled())
      return fail(NTHIP_ERR_UNSUPPORTED, "NTHIP_OUT_READ_SLOTS on fixed-length reads: reads back to back (stride == length) of at most 2048 bases");
    if (flen >= k && rd->n_reads * (uint64_t)(flen - k + 1) > out->capacity) {
      if (total_out) *total_out = rd->n_reads * (uint64_t)(flen - k + 1);
      return fail(NTHIP_ERR_CAPACITY, "output capacity %llu k-mers < %llu needed (the slot array)", (unsigned long long)out->capacity,
                  (unsigned long long)(rd->n_reads * (uint64_t)(flen - k + 1)));
    }
  }
  if ((flags & NTHIP_OUT_READ_SLOTS) && !fslots) { // one pass: read r's k-mers at the slot its length implies (capi_kmer_reads.hip)
    if (flags & (NTHIP_ASYNC | NTHIP_FORCE_GENERAL | NTHIP_FORCE_ROWS))
      return fail(NTHIP_ERR_UNSUPPORTED, "NTHIP_OUT_READ_SLOTS: a plain synchronous call");
    bool handled = false;
    const int rc = run_kmer_reads(c, st, st.offsets, st.offsets + 1, rd->n_reads, total_bytes, k, m, out->capacity, &total,
                                  &handled, nullptr, /*slots*/ true);
    if (total_out) *total_out = total;
    NTCHK(rc);
    if (!handled)
      return fail(NTHIP_ERR_UNSUPPORTED, "NTHIP_OUT_READ_SLOTS: short reads (<= 2048 bases) in order only");
    NTCHK(unstage_outputs(c, out, flags, rd->n_reads, m, total, st));
    HIPCHK(hipStreamSynchronize(c->stream));
    return NTHIP_OK;
  }

  // Offsets: one pass over them on the device (+ one round trip) before anything trusts them -- are they in order
  // and inside the buffer (a decreasing pair would underflow a length), how long is the longest read, and do all
  // reads have one length?  Reads of one length lying back to back (Illumina reads through the offsets API, what
  // nthash::BatchNtHash sends) ARE a fixed-length batch: the fixed-stride kernels hash them 2x faster.
  ReadsShape shape;
  bool have_shape = false;
  if (st.offsets) {
    OffsetsSurvey sv;
    NTCHK(offsets_survey_device(c, st.offsets, rd->n_reads, total_bytes, &sv));
    if (sv.bad) return fail(NTHIP_ERR_ARG, "offsets / spans are not non-decreasing or reach outside the read buffer");
    if (st.pos && sv.max_len > 0xFFFFFFFFull) // (the façade never gets here: it hashes long sequences window by window)
      return fail(NTHIP_ERR_UNSUPPORTED, "out->pos is 32 bits wide: a read of %llu bases cannot report its positions",
                  (unsigned long long)sv.max_len);
    shape.max_len = shape.max_pitch = sv.max_len;
    shape.sum_len = total_bytes >= sv.off0 ? total_bytes - sv.off0 : 0;
    have_shape = true;
    if (sv.uniform && !(flags & (NTHIP_FORCE_GENERAL | NTHIP_ASYNC)) && rd->n_reads >= 1024 && sv.len0 >= 1 &&
        sv.len0 < (1ull << 30) && sv.off0 + rd->n_reads * sv.len0 <= total_bytes) {
      st.seqs += sv.off0;
      st.offsets = nullptr;
      eff.offsets = nullptr;
      eff.fixed_len = (uint32_t)sv.len0;
      eff.stride = 0;
      total_bytes = rd->n_reads * sv.len0;
    }
  }

  const uint32_t len = rd->fixed_len;
  const uint32_t stride = rd->stride ? rd->stride : len;
  uint32_t pad = 0;
  size_t dyn = 0;
  bool done = false;
  // optimistic dense pass: wanted when the dense stream fits the caller's capacity (a batch with non-bases
  // may still fit when the dense stream does not: the counting paths below decide that)
  const bool rows_ok = !rd->offsets && kmer_fixed_eligible(c, len, stride, k, m, &pad, &dyn);
  // (positions do not need the N-aware pass when the batch turns out clean: every window is emitted)
  bool want_fast = !rd->offsets && !(flags & NTHIP_FORCE_GENERAL) && !st.fwd && !st.rev &&
                   !(st.pos && (flags & NTHIP_ASYNC)) &&
                   len >= k && rd->n_reads * (uint64_t)(len - k + 1) <= out->capacity;
  // fixed-length reads that are (or may be) dirty, or whose positions are wanted: N-aware run-split path
  NaPlan na_plan;
  const bool na_ok = !rd->offsets && !(flags & (NTHIP_FORCE_GENERAL | NTHIP_FORCE_ROWS)) &&
                     len >= k && kmer_na_plan(c, len, stride, k, m, st.pos != nullptr, &na_plan);
  // a shape whose last batch held a non-base: no dense pass that gives up at the first N (0.7 ms of 5.7 for 20 M reads)
  const std::array<uint32_t, 4> dirty_key = {len, stride, k, m};
  if (want_fast && na_ok && !c->tune.no_dirty_memory && !(flags & (NTHIP_ASYNC | NTHIP_OUT_READ_SLOTS)) &&
      c->dirty_shapes.count(dirty_key))
    want_fast = false;
  if (!rd->offsets && len < k) {
    // every read shorter than k: nothing is emitted
    if (st.counts) {
      hipLaunchKernelGGL(fill_u64_kernel, dim3(1024), dim3(256), 0, c->stream, st.counts, rd->n_reads, 0ull);
      HIPCHK(hipGetLastError());
    }
    done = true;
  } else if (want_fast) {
    if ((flags & NTHIP_ASYNC) && (flags & (NTHIP_HOST_INPUT | NTHIP_HOST_OUTPUT)))
      return fail(NTHIP_ERR_UNSUPPORTED, "NTHIP_ASYNC takes device-resident buffers");
    const uint32_t nwin = len - k + 1;
    const uint64_t dense = rd->n_reads * (uint64_t)nwin;
    bool fast_ran = true;
    KmerFixedArgs a;
    memset(&a, 0, sizeof a);
    a.seqs = st.seqs;
    a.hashes = st.hashes;
    a.dirty = (uint32_t*)c->d_small;
    a.n_runs = rd->n_reads;
    a.len = len;
    a.stride = stride;
    a.k = k;
    a.m = m;
    a.nwin = nwin;
    a.pad_dwords = pad;
    const uint64_t n_tiles = (rd->n_reads + KF_RUNS_PER_BLOCK - 1) / KF_RUNS_PER_BLOCK;
    if (n_tiles > 0xFFFFFFFFull) return fail(NTHIP_ERR_UNSUPPORTED, "too many reads for one call");
    a.n_tiles = (uint32_t)n_tiles;
    fill_kmer_consts(k, m, a);
    const bool async = (flags & NTHIP_ASYNC) != 0;
    if (!async && c->async_pending)
      return fail(NTHIP_ERR_ARG, "NTHIP_ASYNC batches are pending: call nthip_ctx_take_dirty first");
    if (!c->async_pending) HIPCHK(hipMemsetAsync(c->d_small, 0, 4, c->stream));
    FixedSlots fs;
    if (fslots) NTCHK(kmer_fixed_slots_begin(c, rd->n_reads, total_bytes, &fs));
    int rc;
    RunsPlan plan;
    const bool rows_only = (flags & NTHIP_FORCE_ROWS) != 0;
    GenPlan gplan;
    // the specialised k=31 instantiations (run length 15 / 30 dividing the window count); everything
    // else goes to the general run-split kernel
    const bool planned = !rows_only && !c->tune.no_special && kmer_runs_plan(c, len, stride, k, m, &plan);
    const bool special31 = planned && k == 31 && (plan.C == 15 || (plan.C == 30 && m == 1));
    // (runs per read <= 128: the kernel's division of a run index by it, one multiply and a shift, is exact there)
    const bool any_k = planned && !special31 && !c->tune.no_any_k_runs && plan.rpr <= 128 &&
                       !kmer_runs_chunked_compiled() && // (the windowed experiment build keeps to the k = 31 shapes)
                       kmer_runs_any_k_compiled(k, m, plan.C);
    const bool special = special31 || any_k; // (NTHIP_TUNE_NO_SPECIAL: A/B, the general kernel on these too)
    if (special) {
      // run-split kernel: contiguous write-out (see kmer_runs_kernel.hpp)
      KmerRunsArgs ra;
      memset(&ra, 0, sizeof ra);
      ra.seqs = st.seqs;
      ra.hashes = st.hashes;
      ra.dirty = (uint32_t*)c->d_small;
      ra.vecmap = fs.d_vecmap;
      if (any_k) NTCHK(get_kmer_tab(c, k, &ra.init_tab)); // 4 tables per window word, zero ones past ceil(k/4)
      else NTCHK(get_init_tab(c, k, &ra.init_tab));
      ra.n_reads = rd->n_reads;
      ra.n_runs = rd->n_reads * plan.rpr;
      ra.n_wtiles = (ra.n_runs + 63) / 64;
      ra.len = len;
      ra.stride = stride;
      ra.k = k;
      ra.m = m;
      ra.nwin = nwin;
      ra.C = plan.C;
      ra.rpr = plan.rpr;
      ra.ntab = any_k ? kmer_ntab(k) : (k + 3) / 4;
      ra.waves = plan.waves;
      ra.bits_dwords = plan.bits_dwords;
      ra.tile_u64 = plan.tile_u64;
      ra.inv_rpr = 65536u / plan.rpr + 1u;
      ra.dword_tail = plan.dword_tail;
      ra.ph_tiles = plan.ph_tiles;
      // one tile group per block (set in launch_kmer_runs); NTHIP_TUNE_TILE_MAP overrides for A/B runs
      ra.tile_map = c->tune.has_tile_map ? c->tune.tile_map : 0xFFFFFFFFu;
      memcpy(ra.tab, a.tab, sizeof ra.tab);
      memcpy(ra.mult, a.mult, sizeof ra.mult);
      // NTHIP_TUNE_NO_DWORD_TAIL=1: A/B switch for the slab-tail staging variant (tools/ablate.py)
      const bool dt = plan.dword_tail != 0;
      rc = launch_kmer_runs_special(c, ra, plan, dt);
    } else if (!rows_only && kmer_gen_plan(c, len, stride, k, m, &gplan)) {
      // any other shape: general run-split kernel (kmer_runs_gen_kernel.hpp)
      KmerRunsGenArgs ga;
      const uint4* gen_tab = nullptr;
      NTCHK(get_kmer_tab(c, k, &gen_tab));
      // big batch of a shape not seen before: time the model's run length against the longer ones it may not pick
      // on a slice of the batch (a few launches of ~1 ms), keep the fastest for this context
      const std::array<uint32_t, 4> shape_key = {len, stride, k, m};
      auto tuned = c->run_len_cache.find(shape_key);
      if (tuned == c->run_len_cache.end() && dense >= (1ull << 30) && !async && !c->tune.run_len &&
          !c->tune.run_max && !c->tune.no_autotune) {
        uint32_t cand[6] = {gplan.C, 0, 0, 0, 0, 0};
        const uint32_t caps[3] = {19, 23, 31};
        uint32_t n_cand = 1;
        for (uint32_t cap : caps) {
          GenPlan q;
          if (!kmer_gen_plan(c, len, stride, k, m, &q, false, 0, cap)) continue;
          bool seen = false;
          for (uint32_t i = 0; i < n_cand; ++i) seen = seen || cand[i] == q.C;
          if (!seen) cand[n_cand++] = q.C;
        }
        // ... and against the shorter runs of one and two more runs per read: smaller tiles, more waves in flight (round 6: the
        // reference's benchmark shape -- 100 bp, k = 64, m = 3 -- runs 4.7 % faster on runs of 10 than on the model's 13)
        for (uint32_t more = 1; more <= 2; ++more) {
          const uint32_t rpr = (nwin + gplan.C - 1) / gplan.C + more, cs = (nwin + rpr - 1) / rpr;
          bool seen = cs < 4;
          for (uint32_t i = 0; i < n_cand; ++i) seen = seen || cand[i] == cs;
          if (!seen) cand[n_cand++] = cs;
        }
        uint32_t best_c = gplan.C;
        if (n_cand > 1) {
          nthip_reads slice = *rd;
          // ~512 M k-mers (about a millisecond) per trial: on a quarter of that the candidates' times differed by less
          // than their launch-to-launch spread and a shape could land 10-16 % off its best run length (round 3)
          const uint64_t want = (512ull << 20) / nwin + 1;
          slice.n_reads = rd->n_reads < want ? rd->n_reads : want;
          struct EventPair { // destroyed on every way out of the trials
            hipEvent_t e0 = nullptr, e1 = nullptr;
            ~EventPair()
            {
              if (e0) (void)hipEventDestroy(e0);
              if (e1) (void)hipEventDestroy(e1);
            }
          } ev;
          HIPCHK(hipEventCreate(&ev.e0));
          HIPCHK(hipEventCreate(&ev.e1));
          hipEvent_t e0 = ev.e0, e1 = ev.e1;
          float best_ms = 1e30f;
          bool clean = true;
          for (uint32_t i = 0; i < n_cand && clean; ++i) {
            GenPlan q;
            if (!kmer_gen_plan(c, len, stride, k, m, &q, false, cand[i])) continue;
            fill_gen_args(ga, c, st, &slice, k, m, q, a);
            ga.init_tab = gen_tab;
            ga.vecmap = fs.d_vecmap;
            float ms = 1e30f;
            for (int rep = 0; rep < 3 && clean; ++rep) { // the first launch warms the tables and the clocks; then the better of two
              HIPCHK(hipEventRecord(e0, c->stream));
              const bool prof = c->profiling;
              c->profiling = false;
              const int trc = launch_kmer_gen_dense(c, ga, q.lds, q.nw, q.dword_tail != 0);
              c->profiling = prof;
              NTCHK(trc);
              HIPCHK(hipEventRecord(e1, c->stream));
              HIPCHK(hipMemcpyAsync(c->h_small, c->d_small, 4, hipMemcpyDeviceToHost, c->stream));
              HIPCHK(hipStreamSynchronize(c->stream));
              uint32_t d = 0;
              memcpy(&d, c->h_small, 4);
              if (d) clean = false; // a non-base: the dense kernel stopped early, the times mean nothing
              float t = 1e30f;
              HIPCHK(hipEventElapsedTime(&t, e0, e1));
              if (rep > 0 && t < ms) ms = t;
            }
            // (the model's choice is the first candidate: another one has to beat it by 3 % to replace it)
            if (clean && ms < (i == 0 ? best_ms : 0.97f * best_ms)) { best_ms = ms; best_c = cand[i]; }
          }
          if (clean) tuned = c->run_len_cache.emplace(shape_key, best_c).first;
          else HIPCHK(hipMemsetAsync(c->d_small, 0, 4, c->stream)); // the real pass below finds it again
        } else {
          tuned = c->run_len_cache.emplace(shape_key, gplan.C).first;
        }
      }
      if (tuned != c->run_len_cache.end() && tuned->second != gplan.C &&
          !kmer_gen_plan(c, len, stride, k, m, &gplan, false, tuned->second))
        return fail(NTHIP_ERR_HIP, "run-split plan failed for a tuned run length");
      fill_gen_args(ga, c, st, rd, k, m, gplan, a);
      ga.init_tab = gen_tab;
      ga.vecmap = fs.d_vecmap;
      rc = launch_kmer_gen_dense(c, ga, gplan.lds, gplan.nw, gplan.dword_tail != 0);
    } else if (!rows_ok || fslots) { // (the row-per-read kernel has no read-slots bookkeeping)
      rc = NTHIP_OK;
      fast_ran = false;
    } else rc = launch_kmer_rows(c, a, dyn);
    NTCHK(rc);
    if (fslots && !fast_ran) return fail(NTHIP_ERR_UNSUPPORTED, "NTHIP_OUT_READ_SLOTS: no run-split kernel takes this shape");
    if (async) {
      if (!fast_ran) return fail(NTHIP_ERR_UNSUPPORTED, "NTHIP_ASYNC: no dense kernel takes this shape");
      c->async_pending = true;
      if (st.counts) {
        hipLaunchKernelGGL(fill_u64_kernel, dim3(1024), dim3(256), 0, c->stream, st.counts, rd->n_reads,
                           (uint64_t)nwin);
        HIPCHK(hipGetLastError());
      }
      if (total_out) *total_out = dense;
      return NTHIP_OK;
    }
    uint32_t dirty = 1;
    if (fast_ran) {
      HIPCHK(hipMemcpyAsync(c->h_small, c->d_small, 4, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      memcpy(&dirty, c->h_small, 4);
    }
    if (!dirty) {
      total = dense;
      if (st.counts) {
        hipLaunchKernelGGL(fill_u64_kernel, dim3(1024), dim3(256), 0, c->stream, st.counts, rd->n_reads,
                           (uint64_t)nwin);
        HIPCHK(hipGetLastError());
      }
      if (st.pos && !(fslots && c->pos_listed_only)) { // get_pos() of a read of bases only: the window index
        hipLaunchKernelGGL(fill_window_pos_kernel, dim3(c->n_cu * 8), dim3(256), 0, c->stream, st.pos, rd->n_reads, nwin,
                           (const uint64_t*)nullptr, (const uint64_t*)nullptr);
        HIPCHK(hipGetLastError());
      }
      if (fslots) NTCHK(kmer_fixed_slots_finish(c, st, fs, rd->n_reads, len, stride, k, m, total_bytes, nullptr));
      done = true;
    }
    // dirty: some byte is not ACGTU -> redo on an N-aware path (device side)
  }
  if (!done && (flags & NTHIP_ASYNC))
    return fail(NTHIP_ERR_UNSUPPORTED, "NTHIP_ASYNC: not a plain dense call (offsets, pos / strand outputs, capacity)");
  if (!done && na_ok) {
    KmerFixedArgs consts;
    memset(&consts, 0, sizeof consts);
    fill_kmer_consts(k, m, consts);
    // the run length of the N-aware passes, measured like the dense kernel's (see there): the first big batch of a
    // shape runs count -> scan -> hash on a slice for every candidate
    const uint64_t na_dense = rd->n_reads * (uint64_t)(len - k + 1);
    const std::array<uint32_t, 4> na_key = {len, stride | 0x80000000u, k, m | (st.pos ? 0x100u : 0u)};
    auto na_tuned = c->run_len_cache.find(na_key);
    if (na_tuned == c->run_len_cache.end() && na_dense >= (1ull << 30) && !st.fwd && !st.rev &&
        !c->tune.run_len && !c->tune.run_max && !c->tune.no_autotune) {
      uint32_t cand[4] = {na_plan.g.C, 0, 0, 0}, n_cand = 1;
      for (uint32_t cap : {19u, 23u, 31u}) {
        NaPlan q;
        if (!kmer_na_plan(c, len, stride, k, m, st.pos != nullptr, &q, 0, 0, cap)) continue;
        bool seen = false;
        for (uint32_t i = 0; i < n_cand; ++i) seen = seen || cand[i] == q.g.C;
        if (!seen) cand[n_cand++] = q.g.C;
      }
      uint32_t best_c = na_plan.g.C;
      if (n_cand > 1) {
        nthip_reads slice = *rd;
        const uint64_t want = (256ull << 20) / (len - k + 1) + 1; // wall-clock timing (host round trips inside): longer trials
        slice.n_reads = rd->n_reads < want ? rd->n_reads : want;
        double best_s = 1e30;
        const bool prof = c->profiling;
        c->profiling = false;
        for (uint32_t i = 0; i < n_cand; ++i) {
          NaPlan q;
          if (!kmer_na_plan(c, len, stride, k, m, st.pos != nullptr, &q, 0, cand[i])) continue;
          double sec = 1e30;
          int trc = NTHIP_OK;
          for (int rep = 0; rep < 2 && trc == NTHIP_OK; ++rep) {
            uint64_t tt = 0;
            const auto t0 = std::chrono::steady_clock::now();
            trc = run_kmer_na(c, st, &slice, k, m, q, consts, out->capacity, &tt);
            sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
          }
          if (trc != NTHIP_OK) { // (capacity included: the real pass below reports it, with *total set) no tuning
            best_c = na_plan.g.C;
            break;
          }
          if (sec < (i == 0 ? best_s : 0.96 * best_s)) { best_s = sec; best_c = cand[i]; }
        }
        c->profiling = prof;
      }
      na_tuned = c->run_len_cache.emplace(na_key, best_c).first;
    }
    if (na_tuned != c->run_len_cache.end() && na_tuned->second != na_plan.g.C &&
        !kmer_na_plan(c, len, stride, k, m, st.pos != nullptr, &na_plan, 0, na_tuned->second))
      return fail(NTHIP_ERR_HIP, "N-aware plan failed for a tuned run length");
    // a shape of the specialised kernel: the tiles that lost no window (all but a few) go there, at the compact offsets
    bool special_done = false;
    {
      RunsPlan plan;
      if (!(flags & NTHIP_FORCE_ROWS) && !c->tune.no_special && m == 1 && !st.pos && !st.fwd && !st.rev &&
          kmer_runs_plan(c, len, stride, k, m, &plan)) {
        const bool special31 = k == 31 && (plan.C == 15 || plan.C == 30);
        const bool any_k = !special31 && !c->tune.no_any_k_runs && plan.rpr <= 128 && !kmer_runs_chunked_compiled() &&
                           kmer_runs_any_k_compiled(k, m, plan.C);
        if (special31 || any_k) {
          KmerRunsArgs ra;
          memset(&ra, 0, sizeof ra);
          ra.seqs = st.seqs;
          if (any_k) NTCHK(get_kmer_tab(c, k, &ra.init_tab));
          else NTCHK(get_init_tab(c, k, &ra.init_tab));
          ra.n_reads = rd->n_reads;
          ra.n_runs = rd->n_reads * plan.rpr;
          ra.n_wtiles = (ra.n_runs + 63) / 64;
          ra.len = len;
          ra.stride = stride;
          ra.k = k;
          ra.m = m;
          ra.nwin = len - k + 1;
          ra.C = plan.C;
          ra.rpr = plan.rpr;
          ra.ntab = any_k ? kmer_ntab(k) : (k + 3) / 4;
          ra.waves = plan.waves;
          ra.bits_dwords = plan.bits_dwords;
          ra.inv_rpr = 65536u / plan.rpr + 1u;
          ra.dword_tail = plan.dword_tail;
          ra.ph_tiles = plan.ph_tiles;
          ra.tile_map = c->tune.has_tile_map ? c->tune.tile_map : 0xFFFFFFFFu;
          memcpy(ra.tab, consts.tab, sizeof ra.tab);
          memcpy(ra.mult, consts.mult, sizeof ra.mult);
          int src = run_kmer_na_special(c, st, rd, k, m, plan, ra, plan.dword_tail != 0, consts, out->capacity, &total,
                                        &special_done);
          if (src == NTHIP_ERR_CAPACITY && total_out) *total_out = total;
          NTCHK(src);
        }
      }
    }
    int rc = special_done ? NTHIP_OK : run_kmer_na(c, st, rd, k, m, na_plan, consts, out->capacity, &total);
    if (rc == NTHIP_ERR_CAPACITY && total_out) *total_out = total;
    NTCHK(rc);
    if (total < na_dense) c->dirty_shapes.insert(dirty_key);
    else c->dirty_shapes.erase(dirty_key);
    done = true;
  }
  if (!done && rd->offsets && !(flags & NTHIP_FORCE_GENERAL)) {
    bool handled = false;
    int rc = run_kmer_ragged(c, st, st.offsets, st.offsets + 1, rd->n_reads, total_bytes, k, m, out->capacity,
                             &total, &handled, have_shape ? &shape : nullptr);
    if (rc == NTHIP_ERR_CAPACITY && total_out) *total_out = total;
    NTCHK(rc);
    done = handled;
  }
  if (!done) NTCHK(run_kmer_general(c, st, rd, k, m, out->capacity, &total));
  if (total_out) *total_out = total;
  NTCHK(unstage_outputs(c, out, flags, rd->n_reads, m, total, st));
  HIPCHK(hipStreamSynchronize(c->stream));
  return NTHIP_OK;
}
