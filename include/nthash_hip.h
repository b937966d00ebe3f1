/*
 * include/nthash_hip.h -- the C-ABI of nthash_amd (libnthash_hip.so).
 *
 * This is the drop-in boundary for the ntHash batch hot path on MI355X
 * (gfx950): plain C, plain pointers and sizes, `int` status codes, no C++ or
 * torch types.  The reference (bcgsc/ntHash 2.4.0) has no FFI layer of its
 * own: its boundary is the C++ iterator API of include/nthash/nthash.hpp.
 * Each entry point below names the reference interface whose per-read
 * `while (h.roll()) use(h.hashes())` loop it replaces for a whole batch.  The
 * C++ classes in include/nthash/nthash.hpp (this repo) are implemented on top
 * of this ABI; INTEGRATION.md shows the binding a maintainer of the reference
 * would add.
 *
 * Conventions
 *   - every function returns NTHIP_OK (0) or a negative NTHIP_ERR_* code;
 *     nthip_last_error() returns a thread-local message for the last failure.
 *     Nothing throws, nothing calls exit().
 *   - there is NO CPU fallback: without a usable HIP device every entry point
 *     that touches the device fails with NTHIP_ERR_NODEVICE / NTHIP_ERR_HIP.
 *   - pointers are DEVICE pointers unless the NTHIP_HOST_* flag for that side
 *     is given, in which case the library stages through device memory.
 *   - output order is the reference's: read-major, then emitted position
 *     ascending, then the hash index (NtHash: m values; SeedNtHash: seed-major
 *     n_seeds*m2 values, src/seed.cpp:167-172).
 */
#ifndef NTHASH_HIP_H
#define NTHASH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NTHIP_OK 0
#define NTHIP_ERR_ARG (-1)         /* invalid argument (what the reference raise_error()s on) */
#define NTHIP_ERR_HIP (-2)         /* a HIP runtime call failed */
#define NTHIP_ERR_NODEVICE (-3)    /* no usable gfx950 device */
#define NTHIP_ERR_CAPACITY (-4)    /* out.capacity too small; *total holds the need */
#define NTHIP_ERR_UNSUPPORTED (-5) /* outside the supported domain (e.g. k < 3) */

/* flags */
#define NTHIP_HOST_INPUT 0x1u   /* reads.seqs / reads.offsets are host memory */
#define NTHIP_HOST_OUTPUT 0x2u  /* every non-NULL pointer in nthip_out is host memory */
#define NTHIP_FORCE_GENERAL 0x4u /* skip the fixed-length fast kernels (testing / A-B) */
#define NTHIP_FORCE_ROWS 0x8u    /* use the row-per-read fixed-length kernel (testing / A-B) */
#define NTHIP_ASYNC 0x10u        /* nthip_kmer_hash on device-resident fixed-length reads: launch the dense pass and
                                    return without synchronising (many small batches back to back: the call costs a
                                    kernel launch, not a round trip).  The stream is only valid if no batch since the
                                    last nthip_ctx_take_dirty() held a non-base: call it when the results are needed
                                    (it synchronises) and redo the batches without this flag if it reports one.
                                    NTHIP_ERR_UNSUPPORTED when the call is not a plain dense one (offsets, host
                                    buffers, pos / strand outputs, capacity below n_reads * windows). */

#define NTHIP_PACKED_INPUT 0x20u  /* nthip_kmer_hash: reads.seqs is the packed buffer nthip_pack_reads made (device memory,
                                    fixed-length reads: fixed_len / stride in bases as before, offsets NULL) */
#define NTHIP_PACKED_CLEAN 0x40u  /* with NTHIP_PACKED_INPUT: nthip_pack_reads reported no invalid byte (or none inside any
                                    read), every window is emitted and the validity stream is not read at all */

#define NTHIP_OUT_READ_SLOTS 0x80u /* nthip_kmer_hash with offsets, nthip_kmer_hash_spans (short reads in order): a ONE-PASS
                                    output contract.  Read r's k-mers start at slot_off[r] = sum over the reads before it of
                                    max(len - k + 1, 0) -- a place that depends on the read LENGTHS alone, so nothing has to be
                                    counted before anything is written (no pass that marks the reads with non-bases, no
                                    compaction) -- out->counts[r] (required) says how many of the slot's entries are k-mers:
                                    they come first, in the reference's order; the rest of the slot is zero.  *total = the
                                    extent of the slot array (every window of every read), which out->capacity must hold.
                                    Reads without non-bases -- nearly all of them -- fill their slots: the stream is the
                                    compact one exactly when every count equals its window count.
                                    Fixed-length reads too (offsets == NULL, stride == length <= 2048): slot r is
                                    r * (len - k + 1); the dense kernels run as if the batch were clean and the reads with a
                                    non-base are redone in their slots -- a batch with an N here and there at the speed of
                                    a clean one. */

typedef struct nthip_ctx nthip_ctx;     /* one device + one stream + scratch */
typedef struct nthip_seeds nthip_seeds; /* parsed spaced-seed set (device tables) */

/* A batch of reads: the `const char* seq, size_t seq_len` pair of
 * NtHash::NtHash (include/nthash/nthash.hpp:74-78), once per read. */
typedef struct {
  const char* seqs;        /* concatenated ASCII bases, no separators */
  const uint64_t* offsets; /* n_reads+1 byte offsets into seqs; NULL if fixed_len != 0 */
  uint64_t n_reads;
  uint32_t fixed_len;      /* != 0: read r is seqs[r*stride, r*stride + fixed_len) */
  uint32_t stride;         /* bytes between read starts; 0 means fixed_len.  stride <
                              fixed_len describes overlapping runs of ONE long sequence:
                              stride = R, fixed_len = R + k - 1 hashes R windows per run;
                              stride > fixed_len: padded rows (one read per line of a text
                              file: stride = fixed_len + 1), the padding bytes are never hashed */
} nthip_reads;

/* Where the hash stream goes.  Replaces NtHash::hashes() / get_pos() /
 * get_forward_hash() / get_reverse_hash() (nthash.hpp:163-194) for every
 * emitted k-mer of the batch. */
typedef struct {
  uint64_t* hashes;  /* required: capacity * hashes_per_kmer values */
  uint64_t capacity; /* in k-mers */
  uint64_t* counts;  /* optional: n_reads values, emitted k-mers per read */
  uint32_t* pos;     /* optional: capacity values, get_pos() of each emitted k-mer (32 bits: with offsets, a read of
                        2^32 bases or more is refused with NTHIP_ERR_UNSUPPORTED when positions are asked for) */
  uint64_t* fwd;     /* optional: forward-strand hash(es) of each emitted k-mer: capacity values
                        (k-mer hashing) or capacity * n_seeds values, seed-minor (seed hashing) */
  uint64_t* rev;     /* optional: reverse-strand hash(es), same layout */
} nthip_out;

/* ---- library / context -------------------------------------------------- */
const char* nthip_version(void);
const char* nthip_last_error(void);
int nthip_device_count(int* count);
int nthip_ctx_create(int device, nthip_ctx** ctx);
int nthip_ctx_destroy(nthip_ctx* ctx);
/* release what the context keeps between calls to make them cheap: the pinned and device buffers of the
 * FASTQ / FASTA streaming driver (2 x chunk_bytes pinned + about 6 x chunk_bytes of HBM) and the scan scratch */
int nthip_ctx_trim(nthip_ctx* ctx);
/* The consumers (Bloom / counting-sketch / spaced-seed insert and query) work in rounds whose device scratch -- the hash
 * stream of a round, its lists, its answers -- the context KEEPS between calls (allocating tens of GB anew costs seconds).
 * nthip_ctx_set_scratch_limit bounds what a round may plan with and the context may keep: rounds shrink to fit, what is
 * held over a new limit is released at once.  bytes = 0 restores the default, half of the device's memory; limits
 * under 256 MiB are refused.  nthip_ctx_scratch_info reports what the context holds now and the limit in force. */
int nthip_ctx_set_scratch_limit(nthip_ctx* ctx, size_t bytes);
int nthip_ctx_scratch_info(nthip_ctx* ctx, size_t* kept_bytes, size_t* limit_bytes);
/* run on a caller-owned hipStream_t (e.g. torch's current stream); NULL
 * restores the context's own stream */
int nthip_ctx_set_stream(nthip_ctx* ctx, void* hip_stream);
int nthip_ctx_synchronize(nthip_ctx* ctx);
/* synchronise, report (and clear) whether any NTHIP_ASYNC call since the last one met a byte that is not ACGTU */
int nthip_ctx_take_dirty(nthip_ctx* ctx, int* dirty);
/* when on, every hash call brackets its dominant kernel with HIP events on the
 * launch stream; nthip_last_kernel_ms reads the last bracket (synchronises) */
int nthip_ctx_set_profiling(nthip_ctx* ctx, int on);
int nthip_last_kernel_ms(nthip_ctx* ctx, float* ms, const char** kernel_name);
/* the NTHIP_TUNE_* environment knobs of the measurement tools are read once, when the context is created;
 * this reads them again (A/B runs that change them between variants) and forgets the measured run lengths */
int nthip_ctx_reload_tuning(nthip_ctx* ctx);

/* ---- device memory helpers (for callers without their own allocator) ---- */
int nthip_malloc(nthip_ctx* ctx, size_t bytes, void** dptr);
int nthip_free(nthip_ctx* ctx, void* dptr);
int nthip_memcpy_h2d(nthip_ctx* ctx, void* dst, const void* src, size_t bytes);
int nthip_memcpy_d2h(nthip_ctx* ctx, void* dst, const void* src, size_t bytes);
int nthip_memset(nthip_ctx* ctx, void* d_dst, int byte_value, size_t bytes);
/* Page-locked host memory: buffers handed to the NTHIP_HOST_INPUT / NTHIP_HOST_OUTPUT calls are copied by DMA at PCIe
 * rate when they come from here, through the runtime's pageable path (a third of that, and a CPU core) otherwise.
 * No context: any thread, any device.  The C++ facade's stream windows live in such buffers. */
int nthip_host_alloc(size_t bytes, void** hptr);
int nthip_host_free(void* hptr);

/* ---- the hot path -------------------------------------------------------- */
/*
 * nthip_kmer_hash: for every read r, what
 *     nthash::NtHash h(seq_r, len_r, m, k);            // src/kmer.cpp:200-226
 *     while (h.roll()) emit(h.hashes()[0..m));          // src/kmer.cpp:246-264
 * produces, i.e. the m canonical hashes (src/internal.hpp:104-118) of every
 * window whose k characters are all in ACGTUacgtu, in position order.  Reads
 * shorter than k emit nothing (the iterator would raise_error,
 * src/kmer.cpp:215-219).  *total = number of k-mers emitted.
 * Supported domain: 3 <= k <= 65535 (the reference is undefined for k < 3,
 * SURVEY.md App. B Q7), 1 <= m <= 255.
 */
int nthip_kmer_hash(nthip_ctx* ctx, const nthip_reads* reads, uint16_t k, uint8_t m,
                    const nthip_out* out, uint64_t* total, uint32_t flags);

/*
 * Packed input.  The reference turns bases into 2-bit codes on every call (CONVERT_TAB / RC_CONVERT_TAB,
 * src/internal.hpp:350-418, index its di- / tri- / tetramer tables, src/kmer.cpp:43-73); a pipeline that hashes the
 * same reads at several k (ntCard, multi-k assembly) repeats that per k, and on the GPU the ASCII bytes are 13.5 % of the
 * k = 31 kernel's HBM traffic.  nthip_pack_reads does the step once: the bytes of the batch -- [0, total bytes) of
 * reads->seqs, whatever the read layout -- become a 2-bit code stream (0.25 bytes per base; code = (c >> 1) & 3, base i
 * at bits 2i, 2i + 1) followed by a validity stream (1 bit per base, set = not one of ACGTUacgtu).  d_packed: device
 * memory, 16-byte aligned, nthip_packed_size(total bytes) bytes.  *n_invalid = bytes of the batch that are not bases.
 * nthip_kmer_hash(..., NTHIP_PACKED_INPUT [| NTHIP_PACKED_CLEAN]) then takes reads->seqs = d_packed for the same
 * fixed-length reads (reads->offsets == NULL) and returns exactly what it returns for the ASCII batch (hashes, counts,
 * pos; no strand outputs), for any k and m: clean batches read 0.25 B per base instead of 1, batches with non-bases take
 * the N-aware passes with the validity stream in place of the byte test.
 */
int nthip_packed_size(uint64_t n_bases, size_t* total_bytes, size_t* invalid_offset /* optional */);
int nthip_pack_reads(nthip_ctx* ctx, const nthip_reads* reads, void* d_packed, uint64_t* n_invalid, uint32_t flags);

/*
 * Spaced seeds.  nthip_seeds_create does once what every SeedNtHash constructor
 * redoes per object (check_seeds + get_blocks, src/seed.cpp:19-104, 466-470).
 * *asymmetric (optional) is set if any seed is not a palindrome, the condition
 * the reference warns about (src/seed.cpp:96-102).
 */
int nthip_seeds_create(nthip_ctx* ctx, const char* const* seeds, uint32_t n_seeds, uint16_t k,
                       nthip_seeds** out, int* asymmetric);
int nthip_seeds_destroy(nthip_seeds* seeds);
/*
 * The kernel specialisation cache: for large dense batches nthip_seed_hash compiles a kernel for the
 * very seed set and read length at run time (hiprtc, on a thread of its own: until the code object is there the precompiled
 * kernels hash; a second or two once per seed set and shape, kept in
 * $NTHIP_JIT_CACHE / $XDG_CACHE_HOME/nthash_amd / ~/.cache/nthash_amd afterwards; NTHIP_SEED_JIT=0: never, =1: for every
 * batch the kernel takes).  Without hiprtc the precompiled kernels hash the batch: same stream.
 * nthip_seed_jit_source: the text that is compiled for `seeds` on reads of `len` bases (*out is malloc'ed: free() it;
 * NTHIP_ERR_UNSUPPORTED: a shape without a specialised kernel).  Needs no device.
 */
int nthip_seed_jit_source(const char* const* seeds, uint32_t n_seeds, uint16_t k, uint32_t len, uint8_t m2, char** out);
/*
 * nthip_seed_hash: for every read r, what
 *     nthash::SeedNtHash h(seq_r, len_r, seeds, m2, k); // src/seed.cpp:449-471
 *     while (h.roll()) emit(h.hashes()[0..n_seeds*m2)); // src/seed.cpp:518-544
 * produces, including the reference's position state machine on reads with
 * non-ACGTU characters.  out->fwd / out->rev (optional)
 * receive SeedNtHash::get_forward_hash() / get_reverse_hash(): n_seeds values per k-mer.
 */
int nthip_seed_hash(nthip_ctx* ctx, const nthip_reads* reads, const nthip_seeds* seeds,
                    uint8_t m2, const nthip_out* out, uint64_t* total, uint32_t flags);

/*
 * nthip_kmer_extend: the batched form of the de Bruijn graph query BlindNtHash exists
 * for (include/nthash/nthash.hpp:36-41): for each of n k-mers (n*k ASCII bytes, no
 * separators) the hashes of its 4 possible successors and/or predecessors, i.e. what
 *     nthash::BlindNtHash h(kmer, m, k);                  // src/kmer.cpp:338-353
 *     h.peek("ACGT"[b]);  / h.peek_back("ACGT"[b]);       // src/kmer.cpp:377-393
 * leave in h.hashes().  Layout: next[(i*4 + b)*m + j], prev[(i*4 + b)*m + j], self[i*m + j]
 * (the k-mer's own hashes); any of the three may be NULL.  Like BlindNtHash, bytes are not
 * validated; k-mers must consist of bases (ACGTU, either case) -- for other bytes the
 * reference's constructor reads its tetramer tables out of contract (src/kmer.cpp:50-54).
 */
int nthip_kmer_extend(nthip_ctx* ctx, const char* kmers, uint64_t n_kmers, uint16_t k, uint8_t m,
                      uint64_t* self, uint64_t* next, uint64_t* prev, uint32_t flags);

/*
 * nthip_seed_extend: the same query through spaced seeds -- for each of n windows of k bases (k = the seed set's) the
 * n_seeds*m2 values of its 4 successors and / or predecessors, i.e. what
 *     nthash::BlindSeedNtHash h(kmer, seeds, m2, k);          // src/seed.cpp:666-699
 *     h.roll("ACGT"[b]);  /  h.roll_back("ACGT"[b]);          // src/seed.cpp:701-737
 * leave in h.hashes(), each from a fresh object.  Layout: next[(i*4 + b)*n_seeds*m2 + s*m2 + j], prev likewise,
 * self[i*n_seeds*m2 + s*m2 + j] (the window's own values); any of the three may be NULL.  roll_back() reads a seed's
 * MONOMERS (care runs of one position) from the window it leaves (src/seed.cpp:195-198 reused backwards): reproduced, so
 * prev is what the reference returns, not the masked formula of the predecessor, whenever a seed has monomers.  Bytes are
 * not validated (as BlindSeedNtHash): windows must consist of bases.  Seeds of at most 128 bases.
 */
int nthip_seed_extend(nthip_ctx* ctx, const char* kmers, uint64_t n_kmers, const nthip_seeds* seeds, uint8_t m2,
                      uint64_t* self, uint64_t* next, uint64_t* prev, uint32_t flags);

/*
 * Fused consumers of the k-mer hash stream: what ntHash's callers do with
 * hashes() -- Bloom filter insert / membership (the reference points at btllib's Bloom filters,
 * include/nthash/nthash.hpp:14-17,56-57) -- done inside the hashing kernel, so the 8*m bytes per
 * k-mer that bound nthip_kmer_hash are never written.  For every k-mer NtHash would emit for read r
 * (same emission rule as nthip_kmer_hash) and every i < m, bit  hashes()[i] mod n_bits  of the
 * filter is set (insert) or tested (query).  Filter layout: bit p = bit (p & 7) of byte p >> 3
 * (a plain byte array as in btllib::BloomFilter); DEVICE memory, 4-byte aligned, ceil(n_bits/32)*4
 * bytes, persistent across calls -- build it with several insert calls, test with query.
 *
 * insert: *total (optional) = k-mers consumed.
 * query : hits[r] (optional; host memory with NTHIP_HOST_OUTPUT) = k-mers of read r whose m bits
 *         are all set; *total = k-mers tested, *total_hits = sum of hits.
 * A large batch of fixed-length device-resident reads against a filter beyond the caches is not answered load by load (a
 * 128-byte line per k-mer) but region by region, as the insert goes (the binned query: DESIGN 4.8); same results.
 * Any k >= 3 and m >= 1.  Fixed-length reads (reads->offsets == NULL): fused, any batch size.  Reads of any lengths
 * (reads->offsets: a FASTQ batch): the batch's compact hash stream is produced in ONE round in device scratch and the
 * stream forms consume it -- NTHIP_ERR_UNSUPPORTED when that stream (8 m bytes per base at most) does not fit the
 * device: split the batch.
 */
int nthip_kmer_bloom_insert(nthip_ctx* ctx, const nthip_reads* reads, uint16_t k, uint8_t m,
                            uint8_t* d_filter, uint64_t n_bits, uint64_t* total, uint32_t flags);
int nthip_kmer_bloom_query(nthip_ctx* ctx, const nthip_reads* reads, uint16_t k, uint8_t m,
                           const uint8_t* d_filter, uint64_t n_bits, uint64_t* hits, uint64_t* total,
                           uint64_t* total_hits, uint32_t flags);
/*
 * nthip_seed_bloom_insert / _query: the same filter fed by SPACED SEEDS -- for every window SeedNtHash emits for read r
 * (nthip_seed_hash: the reference's position state machine on reads with non-bases, src/seed.cpp:493-544) every one of its
 * n_seeds * m2 hashes (seed-major, include/nthash/nthash.hpp:313-326, 460-479) sets / tests bit  h mod n_bits.  *total =
 * windows consumed; query: hits[r] = windows of read r whose n_seeds * m2 bits are ALL set (device memory; host memory
 * with NTHIP_HOST_OUTPUT), *total_hits their sum.  Fixed-length reads in rounds, reads given by offsets in rounds;
 * NTHIP_HOST_INPUT is honoured.  The seed hashes of a round go through device scratch (48 B per window for two seeds of
 * three hashes) and the stream forms of the filter: binned insert, per-read query.
 */
int nthip_seed_bloom_insert(nthip_ctx* ctx, const nthip_reads* reads, const nthip_seeds* seeds, uint8_t m2, uint8_t* d_filter,
                            uint64_t n_bits, uint64_t* total, uint32_t flags);
int nthip_seed_bloom_query(nthip_ctx* ctx, const nthip_reads* reads, const nthip_seeds* seeds, uint8_t m2, const uint8_t* d_filter,
                           uint64_t n_bits, uint64_t* hits, uint64_t* total, uint64_t* total_hits, uint32_t flags);
/*
 * nthip_kmer_minhash: per-read MinHash signatures, the other thing callers do with m hashes per k-mer
 * (sketching: one minimum per hash function).  signatures[r*m + i] = the minimum of hashes()[i] over the
 * k-mers NtHash emits for read r (src/kmer.cpp:228-264 for the emission rule, src/internal.hpp:104-118
 * for hashes()[i]); UINT64_MAX for a read without a valid k-mer.  The hashes stay in registers: the
 * kernel reads the bases and writes 8*m bytes per READ.  signatures: device memory, or host memory with
 * NTHIP_HOST_OUTPUT.  *total (optional) = k-mers consumed.  Any k >= 3, m >= 1; fixed-length reads, or reads of any
 * lengths (offsets) through h[0] of the batch's compact stream in one round, as the Bloom entries.
 */
int nthip_kmer_minhash(nthip_ctx* ctx, const nthip_reads* reads, uint16_t k, uint8_t m,
                       uint64_t* signatures, uint64_t* total, uint32_t flags);
/* the same two operations on an already materialised stream of n_kmers*m hashes (device memory):
 * the unfused baseline, and the consumer for shapes the fused kernels do not take */
int nthip_stream_bloom_insert(nthip_ctx* ctx, const uint64_t* d_hashes, uint64_t n_values,
                              uint8_t* d_filter, uint64_t n_bits);
/* membership of every k-mer of a stream (m values per k-mer, as nthip_kmer_hash writes them): d_flags[i] = 1 when all m bits
 * of k-mer i are set (btllib's contains()), else 0; *found (optional) = the number of ones.  Device memory throughout. */
int nthip_stream_bloom_query(nthip_ctx* ctx, const uint64_t* d_hashes, uint64_t n_kmers, uint8_t m,
                             const uint8_t* d_filter, uint64_t n_bits, uint8_t* d_flags, uint64_t* found);

/* k-mer counting sketch (count-min, one-byte counters): every hash value of the batch -- m per k-mer NtHash emits -- adds
 * one to counter (h mod n_counters), saturating at 255; nthip_stream_count_query gives a k-mer's estimate, the smallest
 * of its m counters (hashes: m per k-mer, as nthip_kmer_hash writes them; d_estimates: one byte per k-mer).  The sketch
 * is a plain byte array on the device, 4-byte aligned, n_counters a multiple of 4.  What the reference's callers do with
 * hashes() next to Bloom filters (reference include/nthash/nthash.hpp:14-17, 56-57); the layout is ours.  The reads entry
 * takes fixed-length reads in rounds or reads of any lengths (offsets) in one round, as the other consumers;
 * NTHIP_HOST_INPUT is honoured. */
int nthip_kmer_count_insert(nthip_ctx* ctx, const nthip_reads* reads, uint16_t k, uint8_t m, uint8_t* d_counters,
                            uint64_t n_counters, uint64_t* total, uint32_t flags);
int nthip_stream_count_insert(nthip_ctx* ctx, const uint64_t* d_hashes, uint64_t n_values, uint8_t* d_counters,
                              uint64_t n_counters);
int nthip_stream_count_query(nthip_ctx* ctx, const uint64_t* d_hashes, uint64_t n_kmers, uint8_t m,
                             const uint8_t* d_counters, uint64_t n_counters, uint8_t* d_estimates);
/* nthip_kmer_count_query: the sketch's read side on the reads themselves -- no hash stream.  estimates (device memory; host
 * memory with NTHIP_HOST_OUTPUT): one byte per WINDOW of the batch, read r's windows at slot_off[r] = sum over the reads
 * before it of max(len - k + 1, 0) (fixed-length reads: r * (len - k + 1)): estimates[slot_off[r] + w] = the smallest of the
 * m counters of the k-mer NtHash emits at position w of read r (get_pos() == w; src/kmer.cpp:228-264 for what is emitted,
 * src/internal.hpp:104-118 for hashes()[i]), 0 for a window it skips.  *total (optional) = k-mers NtHash emits.  A large
 * batch of fixed-length device-resident reads against a large sketch goes region by region (the binned query, as
 * nthip_kmer_bloom_query); everything else through the compact stream of rounds of reads and nthip_stream_count_query. */
int nthip_kmer_count_query(nthip_ctx* ctx, const nthip_reads* reads, uint16_t k, uint8_t m, const uint8_t* d_counters,
                           uint64_t n_counters, uint8_t* estimates, uint64_t* total, uint32_t flags);

/* Per-read (w, k)-minimizers: of every w consecutive window positions of a read, the k-mer NtHash emits there with the
 * smallest canonical hash (hashes()[0]; ties: the leftmost); a k-mer holding a non-base is not a candidate, a window
 * without candidates picks nothing, a read with fewer than w windows is one window.  Output, on the device: the picked
 * k-mers read by read and left to right -- min_hashes[j], min_pos[j] (window position inside the read; may be NULL) --
 * and min_offsets[n_reads + 1], read r's minimizers being [min_offsets[r], min_offsets[r + 1]).  *total = their number;
 * NTHIP_ERR_CAPACITY (with *total set) when capacity is smaller.  Fixed-length reads (any batch size, in rounds) or reads
 * given by offsets (one round: the batch's emitted stream must fit the device's free memory); NTHIP_HOST_INPUT is honoured.
 * What sketching tools on ntHash keep of a read (the reference's hashes() is their input: src/kmer.cpp:246-264). */
int nthip_kmer_minimizers(nthip_ctx* ctx, const nthip_reads* reads, uint16_t k, uint32_t w, uint64_t* d_min_hashes,
                          uint32_t* d_min_pos, uint64_t* d_min_offsets, uint64_t capacity, uint64_t* total, uint32_t flags);

/*
 * FASTQ / FASTA -> device batches, done the GPU way: the raw file bytes are
 * uploaded as they are, record boundaries are found on the device, and the k-mer kernels read the
 * sequence lines where they lie -- no host parser, no packing copy.
 *
 * nthip_kmer_hash_spans: nthip_kmer_hash for reads given as spans [d_starts[r], d_ends[r]) of one
 * device buffer of buf_bytes bytes (sequence lines inside a raw FASTQ chunk; spans may be separated
 * by anything).  Same emission rule, order and outputs (out->hashes/counts/pos; fwd/rev must be NULL)
 * as nthip_kmer_hash with offsets.  All pointers are device pointers unless NTHIP_HOST_OUTPUT is set
 * for the outputs.  Any k >= 3, m >= 1.
 */
int nthip_kmer_hash_spans(nthip_ctx* ctx, const char* d_buf, uint64_t buf_bytes, const uint64_t* d_starts,
                          const uint64_t* d_ends, uint64_t n_reads, uint16_t k, uint8_t m,
                          const nthip_out* out, uint64_t* total, uint32_t flags);

/* nthip_kmer_minimizers for reads given as spans of one device buffer (the sequence lines of a raw FASTQ chunk, as
 * nthip_fastx_index finds them): the same picks, in the order of the spans.  Device pointers throughout; one round. */
int nthip_kmer_minimizers_spans(nthip_ctx* ctx, const char* d_buf, uint64_t buf_bytes, const uint64_t* d_starts,
                                const uint64_t* d_ends, uint64_t n_reads, uint16_t k, uint32_t w,
                                uint64_t* d_min_hashes, uint32_t* d_min_pos, uint64_t* d_min_offsets, uint64_t capacity,
                                uint64_t* total);

/* nthip_seed_hash over spans: as nthip_seed_hash (the reference's position state machine, App. B Q3) */
int nthip_seed_hash_spans(nthip_ctx* ctx, const char* d_buf, uint64_t buf_bytes, const uint64_t* d_starts,
                          const uint64_t* d_ends, uint64_t n_reads, const nthip_seeds* seeds, uint8_t m2,
                          const nthip_out* out, uint64_t* total, uint32_t flags);

#define NTHIP_FASTQ 4u  /* 4-line records: @header / sequence / + / quality            */
#define NTHIP_FASTA 2u  /* 2-line records: >header / sequence (one line per sequence)   */
/*
 * nthip_fastx_index: d_buf[0..n_bytes) is a piece of a FASTQ / single-line FASTA file that begins
 * at a record start.  For every COMPLETE record r in it (all of its lines end with '\n' inside the
 * piece) writes the span of its sequence line to d_starts[r], d_ends[r] (a trailing '\r' is dropped).
 * *n_records = complete records, *consumed = one past the last byte of the last complete record
 * (the bytes after it are the beginning of the next piece), *malformed != 0 when a record does not
 * begin with '@' / '>' or a FASTQ record's third line does not begin with '+'.
 * NTHIP_ERR_CAPACITY (with *n_records set) when there are more than `capacity` records.
 */
int nthip_fastx_index(nthip_ctx* ctx, const char* d_buf, uint64_t n_bytes, uint32_t format,
                      uint64_t* d_starts, uint64_t* d_ends, uint64_t capacity, uint64_t* n_records,
                      uint64_t* consumed, int* malformed);

/*
 * nthip_fasta_compact: multi-line FASTA on the device.  d_raw[0..n_bytes) is a whole FASTA text that
 * begins with '>' (records may span any number of lines; CR, LF and blank lines are dropped).
 * d_seqs (>= n_bytes bytes) receives the sequences back to back, d_offsets[0..*n_records] their
 * offsets -- the layout nthip_kmer_hash takes (reads->seqs / reads->offsets, device pointers).
 * *seq_bytes = d_offsets[*n_records].  NTHIP_ERR_CAPACITY (with *n_records set) when the text has
 * more than `capacity` records (d_offsets must hold capacity + 1 values).
 */
int nthip_fasta_compact(nthip_ctx* ctx, const char* d_raw, uint64_t n_bytes, char* d_seqs,
                        uint64_t* d_offsets, uint64_t capacity, uint64_t* n_records, uint64_t* seq_bytes);

#define NTHIP_FASTA_MULTILINE 1u /* file driver: FASTA whose sequences span lines; the whole file is one batch */
/* one batch of a streamed file: every pointer is DEVICE memory owned by the driver, valid during the callback */
typedef struct nthip_fastx_batch {
  uint64_t n_reads;          /* records of this batch, in file order                              */
  uint64_t n_kmers;          /* k-mers emitted for them                                           */
  const uint64_t* hashes;    /* n_kmers * m, order as nthip_kmer_hash                             */
  const uint64_t* counts;    /* n_reads: k-mers per read                                          */
  const char* raw;           /* the raw bytes the spans refer to                                  */
  const uint64_t* starts;    /* n_reads: sequence line of read r = raw[starts[r] .. ends[r])      */
  const uint64_t* ends;
  uint64_t first_read;       /* index of the batch's first record in the file                     */
  int32_t device;            /* the HIP device the pointers above live on (nthip_multi_fastx_*: batches of one file
                                come from several devices, in file order)                          */
  uint32_t reserved;
} nthip_fastx_batch;
typedef int (*nthip_fastx_fn)(void* user, const nthip_fastx_batch* batch); /* non-zero return stops the stream */
typedef struct nthip_fastx_stats {
  uint64_t file_bytes, reads, kmers, batches;
  double seconds;            /* wall time of the whole call                                        */
  double read_seconds;       /* time the reader threads spent in pread / inflate (overlapped)      */
  double gpu_seconds;        /* index + hash + callback time (overlapped with reads and uploads)   */
} nthip_fastx_stats;
/*
 * nthip_fastx_kmer_hash_file: stream a FASTQ / single-line FASTA file through the hash path.
 * Reader threads pread() pieces of chunk_bytes into pinned buffers; each piece is uploaded on a copy
 * stream while the previous one is indexed and hashed; `fn` sees every batch once, in file order.
 * chunk_bytes == 0 picks 256 MiB.  Records longer than 16 MiB are not supported by the streaming
 * formats; NTHIP_FASTA_MULTILINE (genomes: few, long sequences) loads the whole file into HBM,
 * compacts it with nthip_fasta_compact and calls `fn` once (raw = the compacted sequences,
 * starts/ends = their offsets).
 * A file that begins with the gzip magic (1f 8b; any name; concatenated members -- bgzip, cat a.gz b.gz -- included) is
 * inflated on the host by the system's zlib (libz.so.1, loaded at run time: NTHIP_ERR_UNSUPPORTED if it is absent), by ONE
 * thread that feeds the same pinned ring with chunks of the INFLATED stream; batches, order and results are those of the
 * plain file.  A BGZF file (bgzip: 64 KiB members that carry their sizes) is inflated block-wise by all reader threads
 * instead (crc32 and sizes checked; NTHIP_TUNE_NO_BGZF=1 in the environment: the one-thread path).  stats->file_bytes stays
 * the size on disk; a truncated or corrupt stream is NTHIP_ERR_ARG.  The multi-device
 * driver gives a gzip file to its first device (a deflate stream cannot be cut at record starts without inflating it).
 */
int nthip_fastx_kmer_hash_file(nthip_ctx* ctx, const char* path, uint32_t format, uint16_t k, uint8_t m,
                               uint64_t chunk_bytes, nthip_fastx_fn fn, void* user, nthip_fastx_stats* stats);
/* the same stream through SeedNtHash: batch->hashes holds n_seeds*m2 values per k-mer (NTHIP_FASTQ / NTHIP_FASTA) */
int nthip_fastx_seed_hash_file(nthip_ctx* ctx, const char* path, uint32_t format, const nthip_seeds* seeds,
                               uint8_t m2, uint64_t chunk_bytes, nthip_fastx_fn fn, void* user,
                               nthip_fastx_stats* stats);

/* ---- one node, several GPUs --------------------------------
 * Reads are hashed independently of each other (reference include/nthash/nthash.hpp:196-204: all state is per
 * object), so a HOST-resident batch is cut into contiguous shards of reads, one per device, hashed concurrently by one
 * host thread and one context per device; nothing is exchanged.  The result is exactly what ONE nthip_kmer_hash /
 * nthip_seed_hash call returns for the whole batch (read order, N-skipping, counts, positions).  out->capacity must
 * hold every window of the batch (sum over reads of max(len - k + 1, 0)): shards write in place and only move when
 * reads with non-bases left gaps.  devices == NULL: every visible device; a device may be listed more than once
 * (that is how the path is tested on a one-GPU box). */
typedef struct nthip_multi nthip_multi;
typedef struct nthip_multi_seeds nthip_multi_seeds;
int nthip_multi_create(const int* devices, int n_devices, nthip_multi** out);
int nthip_multi_destroy(nthip_multi* multi);
int nthip_multi_device_count(const nthip_multi* multi, int* n);
int nthip_multi_kmer_hash(nthip_multi* multi, const nthip_reads* reads, uint16_t k, uint8_t m, const nthip_out* out,
                          uint64_t* total);
int nthip_multi_seeds_create(nthip_multi* multi, const char* const* seeds, uint32_t n_seeds, uint16_t k,
                             nthip_multi_seeds** out, int* asymmetric);
int nthip_multi_seeds_destroy(nthip_multi_seeds* seeds);
int nthip_multi_seed_hash(nthip_multi* multi, const nthip_reads* reads, const nthip_multi_seeds* seeds, uint8_t m2,
                          const nthip_out* out, uint64_t* total);

/*
 * nthip_multi_fastx_kmer_hash_file: nthip_fastx_kmer_hash_file over the devices of `multi` -- the device-resident
 * multi-GPU path (8 PCIe roots stream 8 x what one does; nthip_multi_kmer_hash above is bound by the hashes coming BACK
 * over PCIe).  The file is cut into pieces of about chunk_bytes that begin and end on record boundaries (found on the
 * host from the line structure), piece j goes to device j mod N; every device runs the single-device pipeline -- its own
 * reader threads, pinned ring, copy stream and context -- on its pieces, nothing is exchanged.  `fn` still sees every
 * batch exactly once and IN FILE ORDER (batch->first_read counts through the file), one call at a time; batch->device
 * says where its pointers live (the callback runs on the worker thread of that device, with that device current).
 * NTHIP_FASTQ / NTHIP_FASTA only.
 */
int nthip_multi_fastx_kmer_hash_file(nthip_multi* multi, const char* path, uint32_t format, uint16_t k, uint8_t m,
                                     uint64_t chunk_bytes, nthip_fastx_fn fn, void* user, nthip_fastx_stats* stats);

/* ---- several GPUs, device-resident (round 4; SURVEY.md 5 "distributed", 8e, 8f-1) ----------------------------------
 * shards[g] (g < the multi's device count): the g-th device's part of the job -- every pointer in it is memory of THAT
 * device (allocate through the context nthip_multi_ctx returns; host memory with NTHIP_HOST_INPUT); n_reads == 0: nothing
 * for that device.  One host thread and one context per device, as nthip_multi_kmer_hash.
 *
 * nthip_multi_kmer_hash_shards: nthip_kmer_hash on every device at once, outs[g] / totals[g] per device (device memory
 * unless NTHIP_HOST_OUTPUT): nothing crosses a link.
 *
 * Consumers: every device consumes its shard into ITS table (filters[g] / counters[g] / sigs[g], device memory of device
 * g, all of one size) exactly as the single-device calls do; then the tables are MERGED over peer copies -- a ring
 * reduce-scatter with the element-wise operator the table needs (OR / saturating add of one-byte counters / minimum of
 * 64-bit entries; hipMemcpyPeerAsync + a fold kernel, RCCL has no such operators) -- so only RESULTS cross xGMI, never
 * a hash stream.  Afterwards tables[0] holds the merged table (what ONE device would have built from all the reads,
 * merged with whatever the tables held before: zero the ones that should contribute nothing); with
 * NTHIP_MULTI_ALLGATHER every tables[g] does (a ring all-gather: what a sharded query needs next).
 *   bloom_insert  n_bits a multiple of 128;   count_insert  n_counters a multiple of 16;
 *   minhash_set   sigs[g]: (m + 1) & ~1 entries of 64 bits; entry i = the minimum of hashes()[i] over EVERY k-mer of every
 *                 read (the m-permutation MinHash signature of the set; the per-read ones: nthip_kmer_minhash)
 * *total (optional): k-mers consumed over all devices.  nthip_multi_merge: the merge alone, on tables built elsewhere.
 * The reference has no counterpart (one object, one thread: include/nthash/nthash.hpp:196-204); the per-device results
 * are those of the single-device calls, pinned there. */
#define NTHIP_MULTI_ALLGATHER 0x100u
#define NTHIP_MERGE_OR 0
#define NTHIP_MERGE_ADD_SAT_U8 1
#define NTHIP_MERGE_MIN_U64 2
int nthip_multi_ctx(nthip_multi* multi, int index, nthip_ctx** ctx);
int nthip_multi_kmer_hash_shards(nthip_multi* multi, const nthip_reads* shards, uint16_t k, uint8_t m, const nthip_out* outs,
                                 uint64_t* totals, uint32_t flags);
int nthip_multi_kmer_bloom_insert(nthip_multi* multi, const nthip_reads* shards, uint16_t k, uint8_t m, uint8_t* const* d_filters,
                                  uint64_t n_bits, uint64_t* total, uint32_t flags);
int nthip_multi_kmer_count_insert(nthip_multi* multi, const nthip_reads* shards, uint16_t k, uint8_t m, uint8_t* const* d_counters,
                                  uint64_t n_counters, uint64_t* total, uint32_t flags);
int nthip_multi_kmer_minhash_set(nthip_multi* multi, const nthip_reads* shards, uint16_t k, uint8_t m, uint64_t* const* d_sigs,
                                 uint64_t* total, uint32_t flags);
int nthip_multi_merge(nthip_multi* multi, void* const* d_tables, uint64_t bytes, int op, uint32_t flags);
/* The sharded QUERY the all-gathered tables are for: reads sharded as above, device g asks ITS copy of the table (filters[g]
 * / counters[g]: what NTHIP_MULTI_ALLGATHER left there, or any table of device g) exactly as nthip_kmer_bloom_query /
 * nthip_kmer_count_query do; hits[g] (may be NULL, or hold NULLs) / estimates[g]: memory of device g (host memory with
 * NTHIP_HOST_OUTPUT), laid out as the single-device calls lay them out for shard g.  *total / *total_hits: sums over the
 * devices.  Nothing crosses a link. */
int nthip_multi_kmer_bloom_query(nthip_multi* multi, const nthip_reads* shards, uint16_t k, uint8_t m, const uint8_t* const* d_filters,
                                 uint64_t n_bits, uint64_t* const* hits, uint64_t* total, uint64_t* total_hits, uint32_t flags);
int nthip_multi_kmer_count_query(nthip_multi* multi, const nthip_reads* shards, uint16_t k, uint8_t m, const uint8_t* const* d_counters,
                                 uint64_t n_counters, uint8_t* const* estimates, uint64_t* total, uint32_t flags);

/* ---- measurement helpers (device-resident synthetic data, checksums) ---- */
/* counter-based reads: read r, 32-base word w ->
 * splitmix64(seed + r*W + w), 2 bits per base, "ACGT"[..]; writes
 * n_reads*len bytes at d_dst */
int nthip_synth_reads(nthip_ctx* ctx, char* d_dst, uint64_t first_read, uint64_t n_reads,
                      uint32_t len, uint64_t seed);
/* wrapping sum and XOR of n device-resident u64 values */
int nthip_checksum(nthip_ctx* ctx, const uint64_t* d_vals, uint64_t n, uint64_t* sum,
                   uint64_t* xr);
/* device-to-device copy rate (the achievable-HBM yardstick next to the 8 TB/s spec) */
int nthip_copy_bench(nthip_ctx* ctx, void* d_dst, const void* d_src, size_t bytes, int reps,
                     float* best_ms);
/* write-only rate: every wave instruction stores one contiguous KiB (the ceiling of a write-bound hash stream) */
int nthip_fill_bench(nthip_ctx* ctx, void* d_dst, size_t bytes, int reps, float* best_ms);
/* Placement-aware allocation for long-lived device buffers (a pipeline's hash stream): which pages hipMalloc hands out
 * decides how fast a buffer streams on MI355X (the same fill: 5.6-7.1 TB/s over fresh allocations; the hash kernels follow
 * it).  Up to `candidates` allocations are made and measured with nthip_fill_bench's write-only pattern, the fastest is
 * returned, the others are released.  The first candidate is a plain hipMalloc; the others are one virtual range mapped
 * from physical pieces of 8-32 MiB (hipMemCreate / hipMemMap): one big physical allocation is the usual way into the slow
 * class, many small ones almost never are -- so the pointer MUST be released with nthip_free of the same context (which
 * knows both kinds), not with hipFree.  It is an ordinary device pointer for kernels and copies of this process.
 * *gbps (optional): its fill rate, *tried (optional): how many were measured.  The buffer's content is undefined.  Buffers
 * under 64 MiB are allocated without a probe. */
int nthip_malloc_probed(nthip_ctx* ctx, size_t bytes, int candidates, void** dptr, double* gbps, int* tried);

#ifdef __cplusplus
}
#endif
#endif /* NTHASH_HIP_H */
