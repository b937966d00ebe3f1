// capi_multi.hip -- one node, several GPUs: host code only.
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
//
// Reads are hashed independently of each other (the reference's iterator keeps all its state in the object,
// include/nthash/nthash.hpp:196-204), so a host-resident batch is cut into contiguous shards of reads, one per device,
// and the shards are hashed concurrently -- one host thread and one context per device, nothing exchanged between
// them.  Each shard writes into the caller's arrays at the place its reads would start if every window were emitted;
// when reads with non-bases left gaps, the later shards are moved down so that the result is exactly the stream a single
// nthip_kmer_hash call returns.
#include "capi_internal.hpp"

#include <algorithm>
#include <thread>

using namespace ntamd;
using namespace ntamd::host;

struct nthip_multi_seeds {
  nthip_multi* owner = nullptr;
  std::vector<nthip_seeds*> per_ctx;
  uint32_t n_seeds = 0, k = 0;
};

extern "C" int nthip_multi_create(const int* devices, int n_devices, nthip_multi** out)
{
  if (!out) return fail(NTHIP_ERR_ARG, "multi out pointer is NULL");
  *out = nullptr;
  std::vector<int> devs;
  if (devices && n_devices > 0) {
    devs.assign(devices, devices + n_devices);
  } else {
    int n = 0;
    NTCHK(nthip_device_count(&n));
    for (int d = 0; d < n; ++d) devs.push_back(d);
  }
  if (devs.empty()) return fail(NTHIP_ERR_NODEVICE, "no HIP device available (nthash_amd has no CPU fallback)");
  nthip_multi* m = new nthip_multi();
  for (int d : devs) {
    nthip_ctx* c = nullptr;
    const int rc = nthip_ctx_create(d, &c);
    if (rc != NTHIP_OK) {
      for (nthip_ctx* x : m->ctx) nthip_ctx_destroy(x);
      delete m;
      return rc;
    }
    m->ctx.push_back(c);
  }
  *out = m;
  return NTHIP_OK;
}

extern "C" int nthip_multi_destroy(nthip_multi* m)
{
  if (!m) return NTHIP_OK;
  for (nthip_ctx* c : m->ctx) nthip_ctx_destroy(c);
  delete m;
  return NTHIP_OK;
}

extern "C" int nthip_multi_device_count(const nthip_multi* m, int* n)
{
  if (!m || !n) return fail(NTHIP_ERR_ARG, "multi / count is NULL");
  *n = (int)m->ctx.size();
  return NTHIP_OK;
}

extern "C" int nthip_multi_seeds_create(nthip_multi* m, const char* const* seeds, uint32_t n_seeds, uint16_t k,
                                        nthip_multi_seeds** out, int* asymmetric)
{
  if (!m || !out) return fail(NTHIP_ERR_ARG, "multi / out is NULL");
  *out = nullptr;
  nthip_multi_seeds* ms = new nthip_multi_seeds();
  ms->owner = m;
  ms->n_seeds = n_seeds;
  ms->k = k;
  for (nthip_ctx* c : m->ctx) {
    nthip_seeds* sd = nullptr;
    const int rc = nthip_seeds_create(c, seeds, n_seeds, k, &sd, asymmetric);
    if (rc != NTHIP_OK) {
      for (nthip_seeds* x : ms->per_ctx) nthip_seeds_destroy(x);
      delete ms;
      return rc;
    }
    ms->per_ctx.push_back(sd);
  }
  *out = ms;
  return NTHIP_OK;
}

extern "C" int nthip_multi_seeds_destroy(nthip_multi_seeds* ms)
{
  if (!ms) return NTHIP_OK;
  for (nthip_seeds* x : ms->per_ctx) nthip_seeds_destroy(x);
  delete ms;
  return NTHIP_OK;
}

namespace {

// windows of read r if every one of them were emitted
inline uint64_t dense_of(const nthip_reads* rd, uint64_t r, uint32_t k)
{
  const uint64_t len = rd->offsets ? rd->offsets[r + 1] - rd->offsets[r] : rd->fixed_len;
  return len >= k ? len - k + 1 : 0;
}

// hash: (device index, shard of reads, shard of outputs) -> status; the shard's emitted count in *tot
template <typename Hash>
int run_sharded(nthip_multi* m, const nthip_reads* rd, uint32_t k, uint32_t per, uint32_t strands_per, const nthip_out* out,
                uint64_t* total_out, Hash hash)
{
  if (total_out) *total_out = 0;
  const uint64_t n = rd->n_reads;
  if (n == 0) return NTHIP_OK;
  const size_t G = m->ctx.size();
  if (rd->offsets)
    for (uint64_t r = 0; r < n; ++r)
      if (rd->offsets[r + 1] < rd->offsets[r])
        return fail(NTHIP_ERR_ARG, "offsets decrease at read %llu", (unsigned long long)r);
  // contiguous shards, balanced by bytes for variable-length reads
  std::vector<uint64_t> first(G + 1, n);
  first[0] = 0;
  for (size_t g = 1; g < G; ++g) {
    if (rd->offsets) {
      const uint64_t want = rd->offsets[0] + (rd->offsets[n] - rd->offsets[0]) / G * g;
      first[g] = (uint64_t)(std::lower_bound(rd->offsets, rd->offsets + n, want) - rd->offsets);
    } else {
      first[g] = n / G * g;
    }
    if (first[g] < first[g - 1]) first[g] = first[g - 1];
  }
  std::vector<uint64_t> dense0(G + 1, 0); // place of a shard's first k-mer if every window were emitted
  for (size_t g = 0; g < G; ++g) {
    uint64_t d = 0;
    if (rd->offsets) {
      for (uint64_t r = first[g]; r < first[g + 1]; ++r) d += dense_of(rd, r, k);
    } else {
      d = (first[g + 1] - first[g]) * dense_of(rd, 0, k);
    }
    dense0[g + 1] = dense0[g] + d;
  }
  if (dense0[G] > out->capacity) {
    if (total_out) *total_out = dense0[G];
    return fail(NTHIP_ERR_CAPACITY, "the multi-device call needs room for every window: capacity %llu < %llu",
                (unsigned long long)out->capacity, (unsigned long long)dense0[G]);
  }
  std::vector<int> rc(G, NTHIP_OK);
  std::vector<uint64_t> tot(G, 0);
  std::vector<std::string> err(G);
  std::vector<std::thread> th;
  for (size_t g = 0; g < G; ++g) {
    th.emplace_back([&, g] {
      const uint64_t r0 = first[g], nr = first[g + 1] - first[g];
      if (nr == 0) return;
      nthip_reads srd = *rd;
      std::vector<uint64_t> offs;
      srd.n_reads = nr;
      if (rd->offsets) {
        offs.resize(nr + 1);
        for (uint64_t i = 0; i <= nr; ++i) offs[i] = rd->offsets[r0 + i] - rd->offsets[r0];
        srd.offsets = offs.data();
        srd.seqs = (const char*)rd->seqs + rd->offsets[r0];
      } else {
        const uint32_t stride = rd->stride ? rd->stride : rd->fixed_len;
        srd.seqs = (const char*)rd->seqs + r0 * (uint64_t)stride;
      }
      nthip_out so = *out;
      so.hashes = out->hashes + dense0[g] * per;
      so.capacity = dense0[g + 1] - dense0[g];
      if (out->counts) so.counts = out->counts + r0;
      if (out->pos) so.pos = out->pos + dense0[g];
      if (out->fwd) so.fwd = out->fwd + dense0[g] * strands_per;
      if (out->rev) so.rev = out->rev + dense0[g] * strands_per;
      rc[g] = hash(g, &srd, &so, &tot[g]);
      if (rc[g] != NTHIP_OK) err[g] = nthip_last_error(); // (thread-local: carry it to the caller's thread)
    });
  }
  for (auto& t : th) t.join();
  for (size_t g = 0; g < G; ++g)
    if (rc[g] != NTHIP_OK) return fail(rc[g], "device shard %zu: %s", g, err[g].c_str());
  // close the gaps reads with non-bases left: shard g moves down to where shard g-1's stream ends
  uint64_t at = tot[0];
  for (size_t g = 1; g < G; ++g) {
    if (tot[g] && at != dense0[g]) {
      memmove(out->hashes + at * per, out->hashes + dense0[g] * per, tot[g] * per * sizeof(uint64_t));
      if (out->pos) memmove(out->pos + at, out->pos + dense0[g], tot[g] * sizeof(uint32_t));
      if (out->fwd) memmove(out->fwd + at * strands_per, out->fwd + dense0[g] * strands_per, tot[g] * strands_per * 8);
      if (out->rev) memmove(out->rev + at * strands_per, out->rev + dense0[g] * strands_per, tot[g] * strands_per * 8);
    }
    at += tot[g];
  }
  if (total_out) *total_out = at;
  return NTHIP_OK;
}

} // namespace

extern "C" int nthip_multi_kmer_hash(nthip_multi* m, const nthip_reads* rd, uint16_t k, uint8_t mh, const nthip_out* out,
                                     uint64_t* total)
{
  if (!m) return fail(NTHIP_ERR_ARG, "multi is NULL");
  NTCHK(check_reads(rd));
  if (!out || !out->hashes) return fail(NTHIP_ERR_ARG, "out->hashes is NULL");
  if (k == 0) return fail(NTHIP_ERR_ARG, "k must be greater than 0");
  if (mh == 0) return fail(NTHIP_ERR_UNSUPPORTED, "num_hashes must be >= 1");
  return run_sharded(m, rd, k, mh, 1, out, total,
                     [&](size_t g, const nthip_reads* srd, const nthip_out* so, uint64_t* tot) {
                       return nthip_kmer_hash(m->ctx[g], srd, k, mh, so, tot, NTHIP_HOST_INPUT | NTHIP_HOST_OUTPUT);
                     });
}

extern "C" int nthip_multi_seed_hash(nthip_multi* m, const nthip_reads* rd, const nthip_multi_seeds* seeds, uint8_t m2,
                                     const nthip_out* out, uint64_t* total)
{
  if (!m || !seeds || seeds->owner != m) return fail(NTHIP_ERR_ARG, "multi / seeds is NULL or of another multi");
  NTCHK(check_reads(rd));
  if (!out || !out->hashes) return fail(NTHIP_ERR_ARG, "out->hashes is NULL");
  if (m2 == 0) return fail(NTHIP_ERR_UNSUPPORTED, "num_hashes_per_seed must be >= 1");
  return run_sharded(m, rd, seeds->k, seeds->n_seeds * m2, seeds->n_seeds, out, total,
                     [&](size_t g, const nthip_reads* srd, const nthip_out* so, uint64_t* tot) {
                       return nthip_seed_hash(m->ctx[g], srd, seeds->per_ctx[g], m2, so, tot,
                                              NTHIP_HOST_INPUT | NTHIP_HOST_OUTPUT);
                     });
}

// ---- the file driver over several devices (device-resident: the hashes stay on the device that made them) --------
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <condition_variable>
#include <mutex>

extern "C" int nthip_multi_fastx_kmer_hash_file(nthip_multi* mm, const char* path, uint32_t format, uint16_t k, uint8_t m,
                                                uint64_t chunk_bytes, nthip_fastx_fn fn, void* user,
                                                nthip_fastx_stats* stats)
{
  if (!mm || !path) return fail(NTHIP_ERR_ARG, "multi / path is NULL");
  if (format != NTHIP_FASTQ && format != NTHIP_FASTA) return fail(NTHIP_ERR_ARG, "format must be NTHIP_FASTQ or NTHIP_FASTA");
  if (k == 0) return fail(NTHIP_ERR_ARG, "k must be greater than 0");
  if (k < 3 || m == 0) return fail(NTHIP_ERR_UNSUPPORTED, "k < 3 / m == 0 are undefined in the reference");
  const auto t_begin = std::chrono::steady_clock::now();
  if (stats) memset(stats, 0, sizeof *stats);
  if (chunk_bytes == 0) chunk_bytes = 256ull << 20;
  if (chunk_bytes < (1ull << 16)) chunk_bytes = 1ull << 16;
  const size_t G = mm->ctx.size();
  // ---- pieces: about chunk_bytes each, cut where a record starts (host: a window of the file per boundary) ----
  std::vector<FastxRange> pieces;
  uint64_t file_size = 0;
  {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(NTHIP_ERR_ARG, "cannot open %s", path);
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); return fail(NTHIP_ERR_ARG, "cannot stat %s", path); }
    file_size = (uint64_t)sb.st_size;
    const int is_gz = file_size ? fastx_file_is_gzip(fd) : 0;
    if (is_gz < 0) { close(fd); return fail(NTHIP_ERR_ARG, "read error on %s", path); }
    if (is_gz) {
      // a deflate stream cannot be cut where records start without inflating it, and one host thread inflates less than
      // one device hashes: the whole file through the first device's pipeline (same batches, same order, same callback)
      close(fd);
      return nthip_fastx_kmer_hash_file(mm->ctx[0], path, format, k, m, chunk_bytes, fn, user, stats);
    }
    uint64_t begin = 0;
    while (begin < file_size) {
      int64_t end = (int64_t)file_size;
      if (begin + chunk_bytes < file_size) {
        // (-2: no boundary could be decided near that place -- a line of a gigabyte --: look one chunk further on rather than
        // hand the whole rest of the file to one device)
        uint64_t pos = begin + chunk_bytes;
        while ((end = fastx_find_record_start(fd, file_size, pos, format)) == -2 && pos + chunk_bytes < file_size) pos += chunk_bytes;
        if (end == -2) end = (int64_t)file_size;
        if (end < 0) { close(fd); return fail(NTHIP_ERR_ARG, "read error on %s", path); }
      }
      if ((uint64_t)end <= begin) end = (int64_t)file_size; // (cannot happen for pos > begin; never loop)
      pieces.push_back({begin, (uint64_t)end - begin});
      begin = (uint64_t)end;
    }
    close(fd);
  }
  if (stats) stats->file_bytes = file_size;
  if (pieces.empty()) return NTHIP_OK;
  // ---- one worker per device: its pieces through its own pipeline; the callback in file order ----
  struct Order {
    std::mutex mu;
    std::condition_variable cv;
    uint64_t next = 0, first_read = 0;
    bool stop = false;
  } order;
  std::vector<std::vector<FastxRange>> mine(G);
  std::vector<std::vector<uint64_t>> index(G);
  for (size_t j = 0; j < pieces.size(); ++j) {
    mine[j % G].push_back(pieces[j]);
    index[j % G].push_back(j);
  }
  std::vector<int> rcs(G, NTHIP_OK);
  std::vector<std::string> errs(G);
  std::vector<nthip_fastx_stats> st(G);
  std::vector<std::thread> workers;
  for (size_t g = 0; g < G; ++g) {
    workers.emplace_back([&, g]() {
      if (mine[g].empty()) return;
      FastxDeliver deliver = [&, g](uint64_t local, nthip_fastx_batch& b) -> int {
        const uint64_t ticket = index[g][local];
        std::unique_lock<std::mutex> lk(order.mu);
        order.cv.wait(lk, [&] { return order.stop || order.next == ticket; });
        if (order.stop) return fail(NTHIP_ERR_ARG, "stopped: another device failed or the callback asked to stop");
        b.first_read = order.first_read;
        int rc = NTHIP_OK;
        if (fn && b.n_reads && fn(user, &b) != 0) rc = fail(NTHIP_ERR_ARG, "stopped by the callback");
        order.first_read += b.n_reads;
        order.next = ticket + 1;
        if (rc != NTHIP_OK) order.stop = true;
        lk.unlock();
        order.cv.notify_all();
        return rc;
      };
      rcs[g] = fastx_stream_ranges(mm->ctx[g], path, format, k, m, nullptr, chunk_bytes, mine[g], deliver, &st[g]);
      if (rcs[g] != NTHIP_OK) {
        errs[g] = nthip_last_error(); // (thread-local: carry it to the caller's thread)
        {
          std::lock_guard<std::mutex> lk(order.mu);
          order.stop = true;
        }
        order.cv.notify_all();
      }
    });
  }
  for (auto& w : workers) w.join();
  for (size_t g = 0; g < G; ++g)
    if (rcs[g] != NTHIP_OK && errs[g].find("stopped: another device") == std::string::npos)
      return fail(rcs[g], "device %d: %s", mm->ctx[g]->device, errs[g].c_str());
  for (size_t g = 0; g < G; ++g)
    if (rcs[g] != NTHIP_OK) return fail(rcs[g], "device %d: %s", mm->ctx[g]->device, errs[g].c_str());
  if (stats) {
    for (size_t g = 0; g < G; ++g) {
      stats->reads += st[g].reads;
      stats->kmers += st[g].kmers;
      stats->batches += st[g].batches;
      stats->read_seconds += st[g].read_seconds;
      stats->gpu_seconds += st[g].gpu_seconds;
    }
    stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
  }
  return NTHIP_OK;
}
