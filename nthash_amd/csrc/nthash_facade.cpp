// nthash_facade.cpp -- libnthash.so: the C++ iterator classes of
// include/nthash/nthash.hpp on top of the C-ABI (include/nthash_hip.h).
//
// Position logic (which window comes next, what is skipped) follows the
// reference's state machines exactly and runs on the host, because it only
// inspects characters.  Hash VALUES for roll() come from the device, one WINDOW
// of the sequence at a time (NTHASH_AMD_WINDOW positions, default 8 Mi: a
// chromosome-sized NtHash needs a few hundred MB of host memory, not 28+8m
// bytes per base of the whole sequence); roll() steps through the window's
// stream and asks for the next window when it runs out.  Every thread has its
// own device context (no lock).
// roll_back()/peek*() and the Blind* classes hash a single caller-chosen base
// per call and evaluate the O(1) recurrence on the host with the shared
// arithmetic of nt_math.hpp.  The same recurrences serve roll() on sequences of
// at most NTHASH_AMD_HOST_ROLL_MAX bases (default 32768): one object per short
// read is the reference's own usage pattern (examples/benchmark.cpp:34-39), and a
// device round trip per object costs 0.2 ms where rolling 100 bases costs under
// a microsecond.  That is a latency decision, not a fallback: the library still
// refuses to construct a hashing object without a HIP device
// (NTHASH_AMD_FORCE_DEVICE=1 sends every sequence to the device; the GPU tests set it),
// and throughput work belongs on nthash::BatchNtHash / the C-ABI, which hash
// many reads per call.
#include "nthash/nthash.hpp"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <functional>
#include <future>
#include <iostream>
#include <map>
#include <mutex>
#include <thread>

#include "nt_math.hpp"
#include "nthash_hip.h"
#include "seed_parse.hpp"

namespace nthash {

using namespace ntamd;

namespace {

// reference: raise_warning / raise_error, src/internal.hpp:9-22
void raise_warning(const std::string& cls, const std::string& msg)
{
  std::cerr << "[ntHash::" << cls << "] \33[33mWARNING: \33[0m" << msg << std::endl;
}
[[noreturn]] void raise_error(const std::string& cls, const std::string& msg)
{
  std::cerr << "[ntHash::" << cls << "] \33[31mERROR: \33[0m" << msg << std::endl;
  std::exit(1);
}

// One device context per thread: objects used from several threads do not serialise on a lock, and a context's
// stream / staging arena are never shared.  Destroyed when the thread ends.
struct ThreadCtx {
  nthip_ctx* ctx = nullptr;
  ~ThreadCtx()
  {
    if (ctx) nthip_ctx_destroy(ctx);
  }
};

// (err: where a thread that must not exit the process -- the helper below -- gets the message instead)
nthip_ctx* device_ctx(const char* cls, std::string* err = nullptr)
{
  static thread_local ThreadCtx tc;
  if (!tc.ctx) {
    int dev = 0;
    if (const char* e = std::getenv("NTHASH_AMD_DEVICE")) dev = std::atoi(e);
    if (nthip_ctx_create(dev, &tc.ctx) != NTHIP_OK) {
      const std::string msg = std::string("GPU hashing unavailable: ") + nthip_last_error();
      if (!err) raise_error(cls, msg);
      *err = msg;
      return nullptr;
    }
  }
  return tc.ctx;
}

// One helper thread per user thread (started by the first long sequence, joined when the user thread ends): it hashes
// window w + 1 of a sequence -- on its own device context and stream -- while the user thread walks window w.
class Helper {
public:
  Helper() : th_([this] { run(); }) {}
  ~Helper()
  {
    {
      std::lock_guard<std::mutex> lock(mu_);
      quit_ = true;
    }
    cv_.notify_all();
    th_.join();
  }
  template <typename F>
  auto submit(F f) -> std::future<decltype(f())>
  {
    auto task = std::make_shared<std::packaged_task<decltype(f())()>>(std::move(f));
    auto fut = task->get_future();
    {
      std::lock_guard<std::mutex> lock(mu_);
      jobs_.emplace_back([task] { (*task)(); });
    }
    cv_.notify_one();
    return fut;
  }

private:
  void run()
  {
    for (;;) {
      std::function<void()> job;
      {
        std::unique_lock<std::mutex> lock(mu_);
        cv_.wait(lock, [&] { return quit_ || !jobs_.empty(); });
        if (jobs_.empty()) return;
        job = std::move(jobs_.front());
        jobs_.pop_front();
      }
      job();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<std::function<void()>> jobs_;
  bool quit_ = false;
  std::thread th_; // (last: the thread starts with everything above in place)
};
Helper& helper()
{
  static thread_local Helper h;
  return h;
}
// NTHASH_AMD_PREFETCH=0: windows one after the other on the user thread (A/B, debugging)
bool prefetch_enabled()
{
  static const bool on = [] {
    const char* e = std::getenv("NTHASH_AMD_PREFETCH");
    return !(e && e[0] == '0');
  }();
  return on;
}

// NTHASH_AMD_DEVICES="all" or "0,1,2,...": nthash::BatchNtHash cuts its batch over these devices (nthip_multi_*).
// Unset: one device (NTHASH_AMD_DEVICE).  One multi-device handle per thread, like the context above.
struct ThreadMulti {
  nthip_multi* m = nullptr;
  bool looked = false;
  ~ThreadMulti()
  {
    if (m) nthip_multi_destroy(m);
  }
};

nthip_multi* device_multi(const char* cls)
{
  static thread_local ThreadMulti tm;
  if (tm.looked) return tm.m;
  tm.looked = true;
  const char* e = std::getenv("NTHASH_AMD_DEVICES");
  if (!e || !*e) return nullptr;
  std::vector<int> devs;
  if (std::string(e) != "all") {
    const char* p = e;
    while (*p) {
      char* end = nullptr;
      const long d = std::strtol(p, &end, 10);
      if (end == p || d < 0) raise_error(cls, std::string("NTHASH_AMD_DEVICES is not \"all\" or a list of device numbers: ") + e);
      devs.push_back((int)d);
      p = (*end == ',') ? end + 1 : end;
      if (*end && *end != ',') raise_error(cls, std::string("NTHASH_AMD_DEVICES is not \"all\" or a list of device numbers: ") + e);
    }
  }
  if (nthip_multi_create(devs.empty() ? nullptr : devs.data(), (int)devs.size(), &tm.m) != NTHIP_OK)
    raise_error(cls, std::string("GPU hashing unavailable: ") + nthip_last_error());
  return tm.m;
}

size_t env_size(const char* name, size_t dflt)
{
  const char* e = std::getenv(name);
  if (!e || !*e) return dflt;
  const unsigned long long v = std::strtoull(e, nullptr, 10);
  return v ? (size_t)v : dflt;
}
// positions per device call of a long sequence
size_t window_positions()
{
  static const size_t w = env_size("NTHASH_AMD_WINDOW", (size_t)8 << 20);
  return w < 1024 ? 1024 : (w > ((size_t)1 << 30) ? ((size_t)1 << 30) : w);
}
// sequences of at most this many bases are rolled on the host (see the file comment); 0 with NTHASH_AMD_FORCE_DEVICE=1
size_t host_roll_max()
{
  static const size_t v = [] {
    const char* f = std::getenv("NTHASH_AMD_FORCE_DEVICE");
    if (f && f[0] == '1') return (size_t)0;
    return env_size("NTHASH_AMD_HOST_ROLL_MAX", 32768);
  }();
  return v;
}

inline bool valid_base(char c) { return is_base((unsigned char)c); }

// The O(1) recurrences (reference src/kmer.cpp:84-114, 164-194).  What a byte adds when it enters or leaves a window
// of k bases is tabulated per k (the reference's SEED_TAB and MS_TAB, src/internal.hpp:121-348, play this role); every
// thread keeps the tables of the last two values of k it used.
} // namespace
namespace detail {
struct RollTables {
  unsigned k = 0;
  uint64_t f_in[256], f_out[256], r_in[256], r_out[256];
};
} // namespace detail
namespace {
using detail::RollTables;
// one immutable table set per k for the life of the process (8 KB each); objects keep the pointer
const RollTables* roll_tables(unsigned k)
{
  static thread_local const RollTables* last = nullptr;
  if (last && last->k == k) return last;
  static std::mutex mu;
  static std::map<unsigned, std::unique_ptr<RollTables>> all;
  std::lock_guard<std::mutex> lock(mu);
  auto& slot = all[k];
  if (!slot) {
    slot.reset(new RollTables());
    RollTables& t = *slot;
    t.k = k;
    for (unsigned c = 0; c < 256; ++c) {
      t.f_in[c] = fwd_seed((unsigned char)c);
      t.f_out[c] = srol_n(t.f_in[c], k);
      t.r_out[c] = rc_seed((unsigned char)c);
      t.r_in[c] = srol_n(t.r_out[c], k);
    }
  }
  last = slot.get();
  return last;
}
inline uint64_t next_fwd(uint64_t f, const RollTables& t, unsigned char out, unsigned char in)
{
  return srol1(f) ^ t.f_in[in] ^ t.f_out[out];
}
inline uint64_t next_rev(uint64_t r, const RollTables& t, unsigned char out, unsigned char in)
{
  return sror1(r ^ t.r_in[in] ^ t.r_out[out]);
}
inline uint64_t prev_fwd(uint64_t f, const RollTables& t, unsigned char out, unsigned char in)
{
  return sror1(f ^ t.f_out[in] ^ t.f_in[out]);
}
inline uint64_t prev_rev(uint64_t r, const RollTables& t, unsigned char out, unsigned char in)
{
  return srol1(r) ^ t.r_out[in] ^ t.r_in[out];
}

// reference: extend_hashes, src/internal.hpp:104-118
inline void extend(uint64_t f, uint64_t r, unsigned k, unsigned m, uint64_t* h)
{
  h[0] = f + r;
  for (unsigned i = 1; i < m; ++i) h[i] = mix_hash(h[0], multiplier(k, i));
}

} // namespace

// ===========================================================================
// device-computed streams
// ===========================================================================
namespace {
// Page-locked host buffers for the windows of the device streams (nthip_host_alloc): the device writes a window's
// positions and hashes there by DMA, nothing is zero-filled or paged in first.  A window's buffer goes back to the pool
// when the last object that walks it lets go; a few are kept for the next window / object / thread, the rest freed.
// (The pool itself is never destroyed: nothing of it may run after the HIP runtime has shut down.)
class PinnedPool {
public:
  void* acquire(size_t bytes, size_t* got, std::string* err)
  {
    {
      std::lock_guard<std::mutex> lock(mu_);
      size_t best = free_.size();
      for (size_t i = 0; i < free_.size(); ++i)
        if (free_[i].second >= bytes && free_[i].second <= 2 * bytes + (1u << 20) &&
            (best == free_.size() || free_[i].second < free_[best].second))
          best = i;
      if (best != free_.size()) {
        void* p = free_[best].first;
        *got = free_[best].second;
        free_.erase(free_.begin() + (std::ptrdiff_t)best);
        return p;
      }
    }
    void* p = nullptr;
    if (nthip_host_alloc(bytes, &p) != NTHIP_OK) {
      *err = std::string("page-locked host memory: ") + nthip_last_error();
      return nullptr;
    }
    *got = bytes;
    return p;
  }
  void release(void* p, size_t bytes)
  {
    {
      std::lock_guard<std::mutex> lock(mu_);
      if (free_.size() < 4) {
        free_.emplace_back(p, bytes);
        return;
      }
    }
    (void)nthip_host_free(p);
  }

private:
  std::mutex mu_;
  std::vector<std::pair<void*, size_t>> free_;
};
PinnedPool& pinned_pool()
{
  static PinnedPool* pool = new PinnedPool();
  return *pool;
}
struct PinnedBlock {
  void* p = nullptr;
  size_t bytes = 0;
  PinnedBlock() = default;
  PinnedBlock(const PinnedBlock&) = delete;
  PinnedBlock& operator=(const PinnedBlock&) = delete;
  ~PinnedBlock()
  {
    if (p) pinned_pool().release(p, bytes);
  }
  // room for n_hashes 64-bit values followed by n_pos 32-bit ones
  bool get(size_t n_hashes, size_t n_pos, uint64_t** hashes, uint32_t** pos, std::string* err)
  {
    const size_t need = n_hashes * sizeof(uint64_t) + n_pos * sizeof(uint32_t) + 64;
    p = pinned_pool().acquire(need, &bytes, err);
    if (!p) return false;
    *hashes = (uint64_t*)p;
    *pos = (uint32_t*)((char*)p + n_hashes * sizeof(uint64_t));
    return true;
  }
};
template <typename T>
struct View { // what a stream needs of std::vector, over memory it does not own
  T* p = nullptr;
  size_t n = 0;
  T* data() const { return p; }
  size_t size() const { return n; }
  T* begin() const { return p; }
  T* end() const { return p + n; }
  T& operator[](size_t i) const { return p[i]; }
};
} // namespace

namespace detail {

// hashes of the windows [w_begin, w_end) of a sequence, as the device returned them
struct KmerStream {
  size_t w_begin = 0, w_end = 0;
  PinnedBlock block;
  View<uint32_t> pos;              // relative to w_begin, ascending
  View<uint64_t> hashes;           // m per entry (the strand hashes stay behind: NtHash::sync_strands)
  unsigned m = 0;
  bool covers(size_t p) const { return p >= w_begin && p < w_end; }
  // index of the entry at (absolute) position p, or npos
  size_t find(size_t p, size_t hint) const
  {
    const uint32_t q = (uint32_t)(p - w_begin);
    if (hint < pos.size() && pos[hint] == q) return hint;
    if (hint + 1 < pos.size() && pos[hint + 1] == q) return hint + 1;
    auto it = std::lower_bound(pos.begin(), pos.end(), q);
    return (it != pos.end() && *it == q) ? (size_t)(it - pos.begin()) : (size_t)-1;
  }
};

struct KmerAhead {
  size_t from = 0;
  std::future<std::shared_ptr<KmerStream>> fut;
  std::shared_ptr<std::string> err; // what went wrong on the helper thread (the user thread raises it)
  ~KmerAhead()
  {
    if (fut.valid()) fut.wait(); // the helper reads the caller's sequence: never outlive the object that borrowed it
  }
};

struct SeedStream;
struct SeedAhead {
  size_t from = 0;
  std::future<std::shared_ptr<SeedStream>> fut;
  std::shared_ptr<std::string> err;
  ~SeedAhead()
  {
    if (fut.valid()) fut.wait(); // the helper reads the caller's sequence: never outlive the object that borrowed it
  }
};

struct SeedSet {
  std::vector<std::string> strings;
  std::vector<SeedShape> shapes;
  unsigned k = 0;
  // the device tables belong to a context, and the helper thread has its own: a second copy for it (round 4 -- until then
  // a SeedNtHash hashed every window on the user thread: 403 M k-mers/s through roll() against NtHash's 1.16 G)
  nthip_seeds* dev_helper = nullptr;
  void* dev_helper_ctx = nullptr;
  // (copies of one SeedNtHash share this set; walked on different threads they share a slot: one call at a time per slot)
  std::mutex mu_user, mu_helper;
  nthip_seeds* dev = nullptr;
  void* dev_ctx = nullptr; // the context `dev` lives in
  ~SeedSet()
  {
    if (dev) nthip_seeds_destroy(dev);
    if (dev_helper) nthip_seeds_destroy(dev_helper);
  }
};

struct SeedStream {
  size_t w_begin = 0, w_end = 0;
  PinnedBlock block;
  View<uint32_t> pos;              // relative to w_begin
  View<uint64_t> hashes;           // n_seeds*m2 per entry (strand hashes: SeedNtHash::sync_strands)
  bool covers(size_t p) const { return p >= w_begin && p < w_end; }
  size_t find(size_t p, size_t hint) const
  {
    const uint32_t q = (uint32_t)(p - w_begin);
    if (hint < pos.size() && pos[hint] == q) return hint;
    if (hint + 1 < pos.size() && pos[hint + 1] == q) return hint + 1;
    auto it = std::lower_bound(pos.begin(), pos.end(), q);
    return (it != pos.end() && *it == q) ? (size_t)(it - pos.begin()) : (size_t)-1;
  }
};

} // namespace detail

namespace {

// the windows [from, from + window_positions()) of the sequence, hashed by one device call on that slice
// (err: failures are reported there instead of ending the process -- the helper thread's calls)
std::shared_ptr<detail::KmerStream> build_kmer_stream(const char* seq, size_t len, unsigned k, unsigned m, size_t from,
                                                      std::string* err = nullptr)
{
  auto st = std::make_shared<detail::KmerStream>();
  st->m = m;
  const size_t n_pos = len - k + 1;
  st->w_begin = from;
  st->w_end = std::min(n_pos, from + window_positions());
  const size_t cap = st->w_end - st->w_begin;
  std::string local_err;
  if (!st->block.get(cap * m, cap, &st->hashes.p, &st->pos.p, err ? err : &local_err)) {
    if (!err) raise_error("NtHash", local_err);
    return nullptr;
  }
  nthip_ctx* ctx = device_ctx("NtHash", err);
  if (!ctx) return nullptr;
  const uint64_t offsets[2] = { 0, (uint64_t)(cap + k - 1) };
  nthip_reads rd = { seq + from, offsets, 1, 0, 0 };
  nthip_out out = { st->hashes.data(), cap, nullptr, st->pos.data(), nullptr, nullptr };
  uint64_t total = 0;
  if (nthip_kmer_hash(ctx, &rd, (uint16_t)k, (uint8_t)m, &out, &total, NTHIP_HOST_INPUT | NTHIP_HOST_OUTPUT) !=
      NTHIP_OK) {
    const std::string msg = std::string("GPU hashing failed: ") + nthip_last_error();
    if (!err) raise_error("NtHash", msg);
    *err = msg;
    return nullptr;
  }
  st->pos.n = total;
  st->hashes.n = total * m;
  return st;
}

// SeedNtHash's emission depends on where the walk started (App. B Q3: a non-base is only acted on when it is the
// INCOMING character of a roll), so a window of the stream is only valid for the walk that reaches its first
// position by rolling.  The device call therefore hashes the slice [from, ...) as a sequence of its own -- exactly
// what the host object does when it (re)initialises at `from` -- and the caller only asks for a window at a position
// where it is about to init() or has just rolled to: see SeedNtHash::set_window.
// (err != nullptr: the helper thread's call -- its own context, the seed set's second copy of the device tables, failures
// reported instead of ending the process)
std::shared_ptr<detail::SeedStream> build_seed_stream(const char* seq, size_t len, size_t from,
                                                      detail::SeedSet& seeds, unsigned m2, std::string* err = nullptr)
{
  auto fail_with = [&](const std::string& msg) -> std::shared_ptr<detail::SeedStream> {
    if (!err) raise_error("SeedNtHash", msg);
    *err = msg;
    return nullptr;
  };
  auto st = std::make_shared<detail::SeedStream>();
  const unsigned k = seeds.k, ns = (unsigned)seeds.strings.size();
  const size_t n_pos = len - k + 1;
  st->w_begin = from;
  st->w_end = std::min(n_pos, from + window_positions());
  const size_t cap = st->w_end - st->w_begin;
  std::string block_err;
  if (!st->block.get(cap * ns * m2, cap, &st->hashes.p, &st->pos.p, &block_err)) return fail_with(block_err);
  nthip_ctx* ctx = device_ctx("SeedNtHash", err);
  if (!ctx) return nullptr;
  std::lock_guard<std::mutex> slot(err ? seeds.mu_helper : seeds.mu_user);
  nthip_seeds*& dev = err ? seeds.dev_helper : seeds.dev;
  void*& dev_ctx = err ? seeds.dev_helper_ctx : seeds.dev_ctx;
  if (!dev || dev_ctx != ctx) { // (the device tables belong to the thread's context)
    if (dev) nthip_seeds_destroy(dev);
    dev = nullptr;
    std::vector<const char*> ptrs;
    for (const auto& s : seeds.strings) ptrs.push_back(s.c_str());
    if (nthip_seeds_create(ctx, ptrs.data(), ns, (uint16_t)k, &dev, nullptr) != NTHIP_OK)
      return fail_with(std::string("GPU seed set-up failed: ") + nthip_last_error());
    dev_ctx = ctx;
  }
  const uint64_t offsets[2] = { 0, (uint64_t)(cap + k - 1) };
  nthip_reads rd = { seq + from, offsets, 1, 0, 0 };
  nthip_out out = { st->hashes.data(), cap, nullptr, st->pos.data(), nullptr, nullptr };
  uint64_t total = 0;
  if (nthip_seed_hash(ctx, &rd, dev, (uint8_t)m2, &out, &total, NTHIP_HOST_INPUT | NTHIP_HOST_OUTPUT) != NTHIP_OK)
    return fail_with(std::string("GPU hashing failed: ") + nthip_last_error());
  st->pos.n = total;
  st->hashes.n = total * ns * m2;
  return st;
}

// Hash of one window for one seed (reference src/seed.cpp:130-207).  The
// reference keeps a blocks-only state and adds the monomers on every call, and
// its backward / peek flavours read the two parts from DIFFERENT windows (see
// the callers), so the block-covered positions and the monomer positions take
// their characters from two accessors.  With blk_at == mono_at this is the
// masked direct formula F = XOR_{p in care} srol^{k-1-p}(S[c_p]).
template <typename BlkAt, typename MonoAt>
void seed_window_hash(const SeedShape& sh, unsigned k, BlkAt blk_at, MonoAt mono_at, uint64_t* f, uint64_t* r)
{
  auto fterm = [&](unsigned p) {
    return (sh.blk_parity[p] ? fwd_seed((unsigned char)blk_at(p)) : 0) ^
           (sh.is_mono[p] ? fwd_seed((unsigned char)mono_at(p)) : 0);
  };
  auto rterm = [&](unsigned p) {
    return (sh.blk_parity[p] ? rc_seed((unsigned char)blk_at(p)) : 0) ^
           (sh.is_mono[p] ? rc_seed((unsigned char)mono_at(p)) : 0);
  };
  uint64_t fh = 0, rh = 0;
  for (unsigned p = 0; p < k; ++p) fh = srol1(fh) ^ fterm(p);
  for (unsigned p = k; p-- > 0;) rh = srol1(rh) ^ rterm(p);
  *f = fh;
  *r = rh;
}

std::shared_ptr<detail::SeedSet> make_seed_set(const std::vector<std::string>& seeds, unsigned k)
{
  auto ss = std::make_shared<detail::SeedSet>();
  ss->k = k;
  ss->strings = seeds;
  for (const auto& s : seeds) ss->shapes.push_back(parse_seed_shape(s));
  return ss;
}

// reference: check_seeds, src/seed.cpp:85-104
void check_seeds(const std::vector<std::string>& seeds, unsigned k)
{
  for (const auto& seed : seeds) {
    if (seed.length() != k)
      raise_error("SeedNtHash", "Spaced seed string length (" + std::to_string(seed.length()) +
                                  ") not equal to k=" + std::to_string(k) + " in " + seed);
    if (!seed_is_symmetric(seed))
      raise_warning("SeedNtHash",
                    "Seed " + seed + " is not symmetric, reverse-complement hashing will be inconsistent");
  }
}

} // namespace

// reference: parse_seeds, src/seed.cpp:431-447
std::vector<std::vector<unsigned>>
parse_seeds(const std::vector<std::string>& seed_strings)
{
  std::vector<std::vector<unsigned>> out;
  for (const auto& s : seed_strings) {
    std::vector<unsigned> dont_care;
    for (unsigned p = 0; p < s.size(); ++p)
      if (s[p] != '1') dont_care.push_back(p);
    out.push_back(dont_care);
  }
  return out;
}

// ===========================================================================
// NtHash
// ===========================================================================
NtHash::NtHash(const char* seq, size_t seq_len, typedefs::NUM_HASHES_TYPE num_hashes, typedefs::K_TYPE k,
               size_t pos)
  : seq_(seq)
  , len_(seq_len)
  , num_hashes_(num_hashes)
  , k_(k)
  , pos_(pos)
  , initialized_(false)
  , hash_arr_(new uint64_t[num_hashes ? num_hashes : 1]())
  , rt_(roll_tables(k))
{
  // reference: src/kmer.cpp:212-225
  if (k == 0) raise_error("NtHash", "k must be greater than 0");
  if (len_ < k)
    raise_error("NtHash", "sequence length (" + std::to_string(len_) + ") is smaller than k (" +
                            std::to_string(k) + ")");
  if (pos > len_ - k)
    raise_error("NtHash", "passed position (" + std::to_string(pos) + ") is larger than sequence length (" +
                            std::to_string(len_) + ")");
}

NtHash::NtHash(const NtHash& o)
  : seq_(o.seq_)
  , len_(o.len_)
  , num_hashes_(o.num_hashes_)
  , k_(o.k_)
  , pos_(o.pos_)
  , initialized_(o.initialized_)
  , fwd_(o.fwd_)
  , rev_(o.rev_)
  , strands_stale_(o.strands_stale_)
  , strands_wanted_(o.strands_wanted_)
  , hash_arr_(new uint64_t[o.num_hashes_ ? o.num_hashes_ : 1])
  , rt_(o.rt_)
  , stream_(o.stream_)
  , cursor_(o.cursor_)
  , sp_(o.sp_)
  , sh_(o.sh_)
  , sn_(o.sn_)
  , sbegin_(o.sbegin_)
{
  std::memcpy(hash_arr_.get(), o.hash_arr_.get(), (num_hashes_ ? num_hashes_ : 1) * sizeof(uint64_t));
}

// (a moved-from object keeps no view into the stream that went with the move: its inline roll() takes the general routine)
NtHash::NtHash(NtHash&& o) noexcept
  : seq_(o.seq_)
  , len_(o.len_)
  , num_hashes_(o.num_hashes_)
  , k_(o.k_)
  , pos_(o.pos_)
  , initialized_(o.initialized_)
  , fwd_(o.fwd_)
  , rev_(o.rev_)
  , strands_stale_(o.strands_stale_)
  , strands_wanted_(o.strands_wanted_)
  , hash_arr_(std::move(o.hash_arr_))
  , rt_(o.rt_)
  , stream_(std::move(o.stream_))
  , cursor_(o.cursor_)
  , ahead_(std::move(o.ahead_))
  , sp_(o.sp_)
  , sh_(o.sh_)
  , sn_(o.sn_)
  , sbegin_(o.sbegin_)
{
  o.sp_ = nullptr;
  o.sh_ = nullptr;
  o.sn_ = 0;
}
NtHash::~NtHash() = default;

// take fwd/rev/hashes of the window at pos_ from the device stream (a short sequence: straight from the bases)
bool NtHash::load_from_stream()
{
  if (len_ <= host_roll_max()) {
    (void)device_ctx("NtHash"); // no HIP device, no hashing object: this is a latency path, not a fallback
    // F = XOR_i srol^{k-1-i}(S[s_i]), R = XOR_i srol^{i}(S[comp s_i]) (src/kmer.cpp:43-73, 123-152), by Horner from the
    // per-byte seed tables; a byte whose seed is 0 is not a base
    const unsigned char* w = (const unsigned char*)seq_ + pos_;
    uint64_t f = 0, r = 0;
    for (size_t i = 0; i < k_; ++i) {
      const uint64_t v = rt_->f_in[w[i]];
      if (!v) return false;
      f = srol1(f) ^ v;
    }
    for (size_t i = k_; i-- > 0;) r = srol1(r) ^ rt_->r_out[w[i]];
    fwd_ = f;
    rev_ = r;
    strands_stale_ = false;
    extend(fwd_, rev_, k_, num_hashes_, hash_arr_.get());
    return true;
  }
  if (!stream_ || !stream_->covers(pos_)) next_stream();
  const size_t i = stream_->find(pos_, cursor_);
  if (i == (size_t)-1) return false;
  cursor_ = i;
  strands_stale_ = true; // (the caller may make them current again: roll())
  std::memcpy(hash_arr_.get(), stream_->hashes.data() + i * num_hashes_, num_hashes_ * sizeof(uint64_t));
  return true;
}

// stream_ := the device stream of the window that holds pos_ -- the one the helper thread has been hashing if the walk
// arrived where it was expected (the end of the previous window, or a skip that stays inside the next one), a call of
// its own otherwise -- and the window after it is put on the helper
void NtHash::next_stream()
{
  // NTHASH_AMD_TRACE=1: what every window change cost the user thread (stderr)
  static const bool trace = [] { const char* e = std::getenv("NTHASH_AMD_TRACE"); return e && e[0] == '1'; }();
  const auto t0 = std::chrono::steady_clock::now();
  bool from_helper = false;
  std::shared_ptr<detail::KmerStream> st;
  if (ahead_) {
    std::shared_ptr<detail::KmerAhead> a = std::move(ahead_);
    ahead_.reset();
    std::shared_ptr<detail::KmerStream> got = a->fut.get();
    if (!got) raise_error("NtHash", *a->err);
    if (got->covers(pos_)) {
      st = std::move(got);
      from_helper = true;
    }
  }
  if (!st) st = build_kmer_stream(seq_, len_, k_, num_hashes_, pos_);
  if (trace)
    std::fprintf(stderr, "[nthash_amd] window at %zu: %s, %.3f ms\n", pos_, from_helper ? "from the helper thread" : "hashed here",
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  stream_ = std::move(st);
  cursor_ = 0;
  sp_ = stream_->pos.data();
  sh_ = stream_->hashes.data();
  sn_ = stream_->pos.size();
  sbegin_ = stream_->w_begin;
  if (prefetch_enabled() && stream_->w_end < len_ - k_ + 1) {
    auto a = std::make_shared<detail::KmerAhead>();
    a->from = stream_->w_end;
    a->err = std::make_shared<std::string>();
    const char* seq = seq_;
    const size_t len = len_, from = a->from;
    const unsigned k = k_, m = num_hashes_;
    std::shared_ptr<std::string> err = a->err;
    a->fut = helper().submit([seq, len, k, m, from, err] { return build_kmer_stream(seq, len, k, m, from, err.get()); });
    ahead_ = std::move(a);
  }
}

// the strand hashes of the window at pos_, straight from its bases (src/kmer.cpp:43-73, 123-152): what
// get_forward_hash() / get_reverse_hash(), roll_back() and peek*() need and the device stream does not carry
void NtHash::sync_strands() const
{
  const unsigned char* w = (const unsigned char*)seq_ + pos_;
  uint64_t f = 0, r = 0;
  for (size_t i = 0; i < k_; ++i) f = srol1(f) ^ rt_->f_in[w[i]];
  for (size_t i = k_; i-- > 0;) r = srol1(r) ^ rt_->r_out[w[i]];
  fwd_ = f;
  rev_ = r;
  strands_stale_ = false;
  strands_wanted_ = true;
}

// reference: NtHash::init, src/kmer.cpp:228-244.  The skip loop only looks at
// characters; index len_ (one past the view, which the reference may read) is
// treated as a terminator, i.e. not a base.
bool NtHash::init()
{
  auto window_invalid = [&](size_t at, size_t& bad) {
    for (size_t i = k_; i-- > 0;) {
      const size_t idx = at + i;
      if (idx >= len_ || rt_->f_in[(unsigned char)seq_[idx]] == 0) { // (seed 0 <=> not ACGTU)
        bad = i;
        return true;
      }
    }
    return false;
  };
  size_t bad = 0;
  while (pos_ <= len_ - k_ + 1 && window_invalid(pos_, bad)) pos_ += bad + 1;
  if (pos_ > len_ - k_) return false;
  if (!load_from_stream())
    raise_error("NtHash", "internal error: valid window missing from the device stream");
  initialized_ = true;
  return true;
}

// (the library still exports NtHash::roll(): the reference's library does, src/kmer.cpp:246)
namespace {
__attribute__((used)) bool (NtHash::*const export_nthash_roll)() = &NtHash::roll;
}

// reference: NtHash::roll, src/kmer.cpp:246-264 (the header's inline roll() is this routine's common case)
bool NtHash::roll_general()
{
  if (!initialized_) return init();
  if (pos_ >= len_ - k_) return false;
  if (!valid_base(seq_[pos_ + k_])) {
    pos_ += k_;
    return init();
  }
  ++pos_;
  if (len_ <= host_roll_max()) { // short sequence: the recurrence (src/kmer.cpp:84-94, 164-174) on the host
    const unsigned char out = (unsigned char)seq_[pos_ - 1], in = (unsigned char)seq_[pos_ + k_ - 1];
    fwd_ = next_fwd(fwd_, *rt_, out, in);
    rev_ = next_rev(rev_, *rt_, out, in);
    extend(fwd_, rev_, k_, num_hashes_, hash_arr_.get());
    return true;
  }
  const bool strands_were_current = !strands_stale_;
  const uint64_t f_before = fwd_, r_before = rev_;
  if (!load_from_stream()) {
    // Only reachable when the object was driven outside the reference's contract
    // (e.g. roll_back() after a failed roll() left pos past the last window, where
    // the reference itself reads out of bounds): the window at pos_ then holds a
    // non-base, so it is not in the stream.  Do what the reference does: roll.
    const unsigned char out = (unsigned char)seq_[pos_ - 1], in = (unsigned char)seq_[pos_ + k_ - 1];
    if (!strands_were_current) { // (the strands of the window we come from)
      --pos_;
      sync_strands();
      ++pos_;
    }
    fwd_ = next_fwd(fwd_, *rt_, out, in);
    rev_ = next_rev(rev_, *rt_, out, in);
    strands_stale_ = false;
    extend(fwd_, rev_, k_, num_hashes_, hash_arr_.get());
  } else if (strands_wanted_ && strands_were_current) {
    // somebody asked for the strand hashes before: keep them current with the O(1) recurrence
    const unsigned char out = (unsigned char)seq_[pos_ - 1], in = (unsigned char)seq_[pos_ + k_ - 1];
    fwd_ = next_fwd(f_before, *rt_, out, in);
    rev_ = next_rev(r_before, *rt_, out, in);
    strands_stale_ = false;
  }
  return true;
}

// reference: NtHash::roll_back, src/kmer.cpp:266-287
bool NtHash::roll_back()
{
  if (!initialized_) return init();
  if (pos_ == 0) return false;
  const char in = seq_[pos_ - 1];
  if (!valid_base(in) && pos_ >= k_) {
    pos_ -= k_;
    return init();
  }
  if (!valid_base(in)) return false;
  if (strands_stale_) sync_strands();
  const unsigned char out = (unsigned char)seq_[pos_ + k_ - 1];
  fwd_ = prev_fwd(fwd_, *rt_, out, (unsigned char)in);
  rev_ = prev_rev(rev_, *rt_, out, (unsigned char)in);
  extend(fwd_, rev_, k_, num_hashes_, hash_arr_.get());
  --pos_;
  return true;
}

// reference: NtHash::peek*, src/kmer.cpp:289-336
bool NtHash::peek()
{
  if (pos_ >= len_ - k_) return false;
  return peek(seq_[pos_ + k_]);
}

bool NtHash::peek(char char_in)
{
  if (!initialized_) return init();
  if (!valid_base(char_in)) return false;
  if (strands_stale_) sync_strands();
  const unsigned char out = (unsigned char)seq_[pos_];
  extend(next_fwd(fwd_, *rt_, out, (unsigned char)char_in), next_rev(rev_, *rt_, out, (unsigned char)char_in), k_,
         num_hashes_, hash_arr_.get());
  return true;
}

bool NtHash::peek_back()
{
  if (pos_ == 0) return false;
  return peek_back(seq_[pos_ - 1]);
}

bool NtHash::peek_back(char char_in)
{
  if (!initialized_) return init();
  if (!valid_base(char_in)) return false;
  if (strands_stale_) sync_strands();
  const unsigned char out = (unsigned char)seq_[pos_ + k_ - 1];
  extend(prev_fwd(fwd_, *rt_, out, (unsigned char)char_in), prev_rev(rev_, *rt_, out, (unsigned char)char_in), k_,
         num_hashes_, hash_arr_.get());
  return true;
}

// ===========================================================================
// BlindNtHash (reference src/kmer.cpp:338-393): one caller-chosen base per call
// ===========================================================================
BlindNtHash::BlindNtHash(const char* seq, typedefs::NUM_HASHES_TYPE num_hashes, typedefs::K_TYPE k, ssize_t pos)
  : window_(seq + pos, seq + pos + k)
  , num_hashes_(num_hashes)
  , pos_(pos)
  , hash_arr_(new uint64_t[num_hashes ? num_hashes : 1]())
  , rt_(roll_tables(k))
{
  if (k == 0) raise_error("BlindNtHash", "k must be greater than 0");
  // the reference hashes seq[0..k) while its window holds seq[pos..pos+k)
  // (src/kmer.cpp:342 vs :350-351) -- reproduced
  fwd_ = direct_fwd(seq, k);
  rev_ = direct_rev(seq, k);
  extend(fwd_, rev_, k, num_hashes_, hash_arr_.get());
}

BlindNtHash::BlindNtHash(const BlindNtHash& o)
  : window_(o.window_)
  , num_hashes_(o.num_hashes_)
  , pos_(o.pos_)
  , fwd_(o.fwd_)
  , rev_(o.rev_)
  , hash_arr_(new uint64_t[o.num_hashes_ ? o.num_hashes_ : 1])
  , rt_(o.rt_)
{
  std::memcpy(hash_arr_.get(), o.hash_arr_.get(), (num_hashes_ ? num_hashes_ : 1) * sizeof(uint64_t));
}

void BlindNtHash::roll(char char_in)
{
  const unsigned k = (unsigned)window_.size();
  fwd_ = next_fwd(fwd_, *rt_, (unsigned char)window_.front(), (unsigned char)char_in);
  rev_ = next_rev(rev_, *rt_, (unsigned char)window_.front(), (unsigned char)char_in);
  extend(fwd_, rev_, k, num_hashes_, hash_arr_.get());
  window_.pop_front();
  window_.push_back(char_in);
  ++pos_;
}

void BlindNtHash::roll_back(char char_in)
{
  const unsigned k = (unsigned)window_.size();
  fwd_ = prev_fwd(fwd_, *rt_, (unsigned char)window_.back(), (unsigned char)char_in);
  rev_ = prev_rev(rev_, *rt_, (unsigned char)window_.back(), (unsigned char)char_in);
  extend(fwd_, rev_, k, num_hashes_, hash_arr_.get());
  window_.pop_back();
  window_.push_front(char_in);
  --pos_;
}

void BlindNtHash::peek(char char_in)
{
  const unsigned k = (unsigned)window_.size();
  extend(next_fwd(fwd_, *rt_, (unsigned char)window_.front(), (unsigned char)char_in),
         next_rev(rev_, *rt_, (unsigned char)window_.front(), (unsigned char)char_in), k, num_hashes_,
         hash_arr_.get());
}

void BlindNtHash::peek_back(char char_in)
{
  const unsigned k = (unsigned)window_.size();
  extend(prev_fwd(fwd_, *rt_, (unsigned char)window_.back(), (unsigned char)char_in),
         prev_rev(rev_, *rt_, (unsigned char)window_.back(), (unsigned char)char_in), k, num_hashes_,
         hash_arr_.get());
}

// ===========================================================================
// SeedNtHash
// ===========================================================================
SeedNtHash::SeedNtHash(const char* seq, size_t seq_len, const std::vector<std::string>& seeds,
                       typedefs::NUM_HASHES_TYPE num_hashes_per_seed, typedefs::K_TYPE k, size_t pos)
  : seq_(seq)
  , len_(seq_len)
  , num_hashes_per_seed_(num_hashes_per_seed)
  , k_(k)
  , pos_(pos)
  , pos0_(pos)
  , initialized_(false)
  , n_seeds_((unsigned)seeds.size())
  , fwd_(new uint64_t[seeds.size() ? seeds.size() : 1]())
  , rev_(new uint64_t[seeds.size() ? seeds.size() : 1]())
  , hash_arr_(new uint64_t[std::max<size_t>(1, (size_t)num_hashes_per_seed * seeds.size())]())
{
  // reference: src/seed.cpp:466-470
  check_seeds(seeds, k);
  if (seeds.empty() || seeds[0].size() != k)
    raise_error("SeedNtHash", "k should be equal to seed string lengths");
  seeds_ = make_seed_set(seeds, k);
}

SeedNtHash::SeedNtHash(const char* seq, size_t seq_len, const std::vector<std::vector<unsigned>>& seeds,
                       typedefs::NUM_HASHES_TYPE num_hashes_per_seed, typedefs::K_TYPE k, size_t pos)
  : seq_(seq)
  , len_(seq_len)
  , num_hashes_per_seed_(num_hashes_per_seed)
  , k_(k)
  , pos_(pos)
  , pos0_(pos)
  , initialized_(false)
  , n_seeds_((unsigned)seeds.size())
  , fwd_(new uint64_t[seeds.size() ? seeds.size() : 1]())
  , rev_(new uint64_t[seeds.size() ? seeds.size() : 1]())
  , hash_arr_(new uint64_t[std::max<size_t>(1, (size_t)num_hashes_per_seed * seeds.size())]())
{
  // reference: parsed_seeds_to_blocks, src/seed.cpp:68-83 (no check_seeds on this path)
  std::vector<std::string> strings;
  for (const auto& dont_care : seeds) {
    std::string s(k, '1');
    for (unsigned p : dont_care) s[p] = '0';
    strings.push_back(s);
  }
  seeds_ = make_seed_set(strings, k);
}

SeedNtHash::SeedNtHash(const SeedNtHash& o)
  : seq_(o.seq_)
  , len_(o.len_)
  , num_hashes_per_seed_(o.num_hashes_per_seed_)
  , k_(o.k_)
  , pos_(o.pos_)
  , pos0_(o.pos0_)
  , strands_stale_(o.strands_stale_)
  , initialized_(o.initialized_)
  , n_seeds_(o.n_seeds_)
  , seeds_(o.seeds_)
  , fwd_(new uint64_t[o.n_seeds_ ? o.n_seeds_ : 1])
  , rev_(new uint64_t[o.n_seeds_ ? o.n_seeds_ : 1])
  , hash_arr_(new uint64_t[o.get_hash_num() ? o.get_hash_num() : 1])
  , stream_(o.stream_)
  , cursor_(o.cursor_)
  , sp_(o.sp_)
  , sh_(o.sh_)
  , sn_(o.sn_)
  , sbegin_(o.sbegin_)
{
  std::memcpy(fwd_.get(), o.fwd_.get(), n_seeds_ * sizeof(uint64_t));
  std::memcpy(rev_.get(), o.rev_.get(), n_seeds_ * sizeof(uint64_t));
  std::memcpy(hash_arr_.get(), o.hash_arr_.get(), get_hash_num() * sizeof(uint64_t));
}

// (a moved-from object keeps no view into the stream that went with the move)
SeedNtHash::SeedNtHash(SeedNtHash&& o) noexcept
  : seq_(o.seq_)
  , len_(o.len_)
  , num_hashes_per_seed_(o.num_hashes_per_seed_)
  , k_(o.k_)
  , pos_(o.pos_)
  , pos0_(o.pos0_)
  , strands_stale_(o.strands_stale_)
  , initialized_(o.initialized_)
  , n_seeds_(o.n_seeds_)
  , seeds_(std::move(o.seeds_))
  , fwd_(std::move(o.fwd_))
  , rev_(std::move(o.rev_))
  , hash_arr_(std::move(o.hash_arr_))
  , stream_(std::move(o.stream_))
  , ahead_(std::move(o.ahead_))
  , cursor_(o.cursor_)
  , sp_(o.sp_)
  , sh_(o.sh_)
  , sn_(o.sn_)
  , sbegin_(o.sbegin_)
{
  o.sp_ = nullptr;
  o.sh_ = nullptr;
  o.sn_ = 0;
}
SeedNtHash::~SeedNtHash() = default;

// hashes of the window `win` (k characters).  A window of the sequence itself
// is taken from the device stream when the stream holds it.
void SeedNtHash::set_window(const char* win, bool try_stream)
{
  if (try_stream && len_ <= host_roll_max()) {
    (void)device_ctx("SeedNtHash"); // no HIP device, no hashing object (a latency path, not a fallback)
    try_stream = false;
  }
  if (try_stream && win >= seq_ && win + k_ <= seq_ + len_) {
    const size_t p = (size_t)(win - seq_);
    // a window of the device stream starts where this walk is (first use: at pos0_): the device then walks the slice
    // the way this object does from here on, and a position its walk does not visit is hashed below
    if (!stream_ || !stream_->covers(p)) {
      // the window the helper thread has been hashing is the one wanted when the walk arrived where it was expected: a
      // window is the slice from `from` on hashed as a sequence of its own, on the helper exactly as here
      const size_t from = stream_ ? p : std::min(p, pos0_);
      std::shared_ptr<detail::SeedStream> st;
      if (ahead_) {
        std::shared_ptr<detail::SeedAhead> a = std::move(ahead_);
        ahead_.reset();
        std::shared_ptr<detail::SeedStream> got = a->fut.get();
        if (!got) raise_error("SeedNtHash", *a->err);
        if (a->from == from) st = std::move(got);
      }
      if (!st) st = build_seed_stream(seq_, len_, from, *seeds_, num_hashes_per_seed_);
      stream_ = std::move(st);
      if (prefetch_enabled() && stream_->w_end < len_ - k_ + 1) {
        auto a = std::make_shared<detail::SeedAhead>();
        a->from = stream_->w_end;
        a->err = std::make_shared<std::string>();
        const char* seq = seq_;
        const size_t len = len_, nxt = a->from;
        const unsigned m2 = num_hashes_per_seed_;
        std::shared_ptr<detail::SeedSet> seeds = seeds_;
        std::shared_ptr<std::string> err = a->err;
        a->fut = helper().submit([seq, len, nxt, seeds, m2, err] { return build_seed_stream(seq, len, nxt, *seeds, m2, err.get()); });
        ahead_ = std::move(a);
      }
      cursor_ = 0;
      sp_ = stream_->pos.data();
      sh_ = stream_->hashes.data();
      sn_ = stream_->pos.size();
      sbegin_ = stream_->w_begin;
    }
    const size_t i = stream_->covers(p) ? stream_->find(p, cursor_) : (size_t)-1;
    if (i != (size_t)-1) {
      cursor_ = i;
      strands_stale_ = true; // (win == seq_ + pos_ here: sync_strands hashes that window when a getter asks)
      std::memcpy(hash_arr_.get(), stream_->hashes.data() + i * get_hash_num(),
                  get_hash_num() * sizeof(uint64_t));
      return;
    }
  }
  for (unsigned s = 0; s < n_seeds_; ++s) {
    auto at = [&](unsigned i) { return win[i]; };
    seed_window_hash(seeds_->shapes[s], k_, at, at, &fwd_[s], &rev_[s]);
    extend(fwd_[s], rev_[s], k_, num_hashes_per_seed_, hash_arr_.get() + (size_t)s * num_hashes_per_seed_);
  }
  strands_stale_ = false;
}

// the per-seed strand hashes of the window at pos_ (what set_window's host branch computes; the device stream carries
// hashes() only)
void SeedNtHash::sync_strands() const
{
  const char* win = seq_ + pos_;
  for (unsigned s = 0; s < n_seeds_; ++s) {
    auto at = [&](unsigned i) { return win[i]; };
    seed_window_hash(seeds_->shapes[s], k_, at, at, &fwd_[s], &rev_[s]);
  }
  strands_stale_ = false;
}

// reference: SeedNtHash::init, src/seed.cpp:493-516 -- the first-window routine
// fails on the first NUL met while walking seeds -> blocks -> positions
bool SeedNtHash::init(bool from_roll)
{
  auto first_nul = [&](const char* win, unsigned& where) {
    for (const auto& sh : seeds_->shapes)
      for (size_t b = 0; b + 1 < sh.block_pairs.size(); b += 2)
        for (uint32_t p = sh.block_pairs[b]; p < sh.block_pairs[b + 1]; ++p)
          if (win[p] == 0) {
            where = p;
            return true;
          }
    return false;
  };
  unsigned where = 0;
  while (pos_ < len_ - k_ + 1 && first_nul(seq_ + pos_, where)) pos_ += where + 1;
  if (pos_ > len_ - k_) return false;
  set_window(seq_ + pos_, from_roll);
  initialized_ = true;
  return true;
}

// (the library still exports SeedNtHash::roll(): the reference's library does, src/seed.cpp:518)
namespace {
__attribute__((used)) bool (SeedNtHash::*const export_seednthash_roll)() = &SeedNtHash::roll;
}

// reference: SeedNtHash::roll, src/seed.cpp:518-544 (the header's inline roll() is this routine's common case)
bool SeedNtHash::roll_general()
{
  if (!initialized_) return init(true);
  if (pos_ >= len_ - k_) return false;
  if (!valid_base(seq_[pos_ + k_])) {
    pos_ += k_;
    return init(true);
  }
  ++pos_;
  set_window(seq_ + pos_, true);
  return true;
}

// reference: SeedNtHash::roll_back, src/seed.cpp:546-575
bool SeedNtHash::roll_back()
{
  if (!initialized_) return init(false);
  if (pos_ == 0) return false;
  if (!valid_base(seq_[pos_ - 1]) && pos_ >= k_) {
    pos_ -= k_;
    return init(false);
  }
  if (!valid_base(seq_[pos_ - 1])) return false;
  --pos_;
  hash_backward(true);
  return true;
}

// The reference's backward flavour (ntmsm64l, src/seed.cpp:293-333) rolls the
// blocks-only state back correctly but then adds the monomers with the forward
// macro's index `kmer_seq[pos + 1]` (src/seed.cpp:195-198), i.e. from the window
// it came FROM.  Block-covered positions therefore see the window at pos_,
// monomer positions the window at pos_ + 1.  Reproduced bit for bit.
void SeedNtHash::hash_backward(bool commit)
{
  const char* nw = seq_ + pos_;
  for (unsigned s = 0; s < n_seeds_; ++s) {
    uint64_t f, r;
    // (an object driven past its last window -- roll_back() after a roll() that failed while skipping -- would read
    // beyond the sequence here, as the reference does: found by the sanitizer job; such bytes read as NUL)
    seed_window_hash(seeds_->shapes[s], k_, [&](unsigned i) { return pos_ + i < len_ ? nw[i] : '\0'; },
                     [&](unsigned i) { return pos_ + i + 1 < len_ ? nw[i + 1] : '\0'; }, &f, &r);
    if (commit) {
      fwd_[s] = f;
      rev_[s] = r;
      strands_stale_ = false;
    }
    extend(f, r, k_, num_hashes_per_seed_, hash_arr_.get() + (size_t)s * num_hashes_per_seed_);
  }
}

// reference: SeedNtHash::peek*, src/seed.cpp:577-667 (state is not advanced;
// unlike NtHash::peek(char), the character is not validated)
bool SeedNtHash::peek()
{
  if (pos_ >= len_ - k_) return false;
  return peek(seq_[pos_ + k_]);
}

// Forward peek (src/seed.cpp:356-379): blocks take char_in at the new last
// position; monomers are read from the sequence itself (`kmer_seq[pos + 1]`),
// so a monomer at position k-1 sees seq[pos+k], not char_in.
bool SeedNtHash::peek(char char_in)
{
  if (!initialized_) return init(false);
  const char* nw = seq_ + pos_ + 1;
  for (unsigned s = 0; s < n_seeds_; ++s) {
    uint64_t f, r;
    seed_window_hash(
      seeds_->shapes[s], k_, [&](unsigned i) { return i + 1 < k_ ? nw[i] : char_in; },
      [&](unsigned i) { return pos_ + 1 + i < len_ ? nw[i] : '\0'; }, &f, &r);
    extend(f, r, k_, num_hashes_per_seed_, hash_arr_.get() + (size_t)s * num_hashes_per_seed_);
  }
  return true;
}

bool SeedNtHash::peek_back()
{
  if (pos_ == 0) return false;
  return peek_back(seq_[pos_ - 1]);
}

// Backward peek (src/seed.cpp:402-425): the guard `i_in > k - 1` copied from the
// forward flavour never fires for i_in = block[0], so char_in is not used at
// all: the result is that of roll_back() on the sequence, state untouched.
bool SeedNtHash::peek_back(char /*char_in*/)
{
  if (!initialized_) return init(false);
  --pos_;
  hash_backward(false);
  ++pos_;
  return true;
}

// ===========================================================================
// BlindSeedNtHash (reference src/seed.cpp:669-737)
// ===========================================================================
BlindSeedNtHash::BlindSeedNtHash(const char* seq, const std::vector<std::string>& seeds,
                                 typedefs::NUM_HASHES_TYPE num_hashes_per_seed, typedefs::K_TYPE k, ssize_t pos)
  : window_(seq + pos, seq + pos + k)
  , num_hashes_per_seed_(num_hashes_per_seed)
  , k_(k)
  , pos_(pos)
  , n_seeds_((unsigned)seeds.size())
  , fwd_(new uint64_t[seeds.size() ? seeds.size() : 1]())
  , rev_(new uint64_t[seeds.size() ? seeds.size() : 1]())
  , hash_arr_(new uint64_t[std::max<size_t>(1, (size_t)num_hashes_per_seed * seeds.size())]())
{
  check_seeds(seeds, k);
  seeds_ = make_seed_set(seeds, k);
  rehash();
}

BlindSeedNtHash::BlindSeedNtHash(const BlindSeedNtHash& o)
  : window_(o.window_)
  , num_hashes_per_seed_(o.num_hashes_per_seed_)
  , k_(o.k_)
  , pos_(o.pos_)
  , n_seeds_(o.n_seeds_)
  , seeds_(o.seeds_)
  , fwd_(new uint64_t[o.n_seeds_ ? o.n_seeds_ : 1])
  , rev_(new uint64_t[o.n_seeds_ ? o.n_seeds_ : 1])
  , hash_arr_(new uint64_t[o.get_hash_num() ? o.get_hash_num() : 1])
{
  std::memcpy(fwd_.get(), o.fwd_.get(), n_seeds_ * sizeof(uint64_t));
  std::memcpy(rev_.get(), o.rev_.get(), n_seeds_ * sizeof(uint64_t));
  std::memcpy(hash_arr_.get(), o.hash_arr_.get(), get_hash_num() * sizeof(uint64_t));
}

// hashes of the current window_ (its first k_ characters)
void BlindSeedNtHash::rehash()
{
  for (unsigned s = 0; s < n_seeds_; ++s) {
    auto at = [&](unsigned i) { return window_[i]; };
    seed_window_hash(seeds_->shapes[s], k_, at, at, &fwd_[s], &rev_[s]);
    extend(fwd_[s], rev_[s], k_, num_hashes_per_seed_, hash_arr_.get() + (size_t)s * num_hashes_per_seed_);
  }
}

void BlindSeedNtHash::roll(char char_in)
{
  window_.push_back(char_in);
  window_.pop_front();
  rehash();
  ++pos_;
}

// backward flavour: monomers from the window being left (see SeedNtHash::hash_backward)
void BlindSeedNtHash::roll_back(char char_in)
{
  window_.push_front(char_in); // k+1 characters: [0,k) new window, [1,k] old window
  for (unsigned s = 0; s < n_seeds_; ++s) {
    seed_window_hash(seeds_->shapes[s], k_, [&](unsigned i) { return window_[i]; },
                     [&](unsigned i) { return window_[i + 1]; }, &fwd_[s], &rev_[s]);
    extend(fwd_[s], rev_[s], k_, num_hashes_per_seed_, hash_arr_.get() + (size_t)s * num_hashes_per_seed_);
  }
  window_.pop_back();
  --pos_;
}

// ===========================================================================
// BatchNtHash: many reads, one device call (nthip_kmer_hash with offsets, counts and positions)
// ===========================================================================
BatchNtHash::BatchNtHash(typedefs::NUM_HASHES_TYPE num_hashes, typedefs::K_TYPE k)
  : num_hashes_(num_hashes)
  , k_(k)
  , offsets_(1, 0)
{
  if (k == 0) raise_error("BatchNtHash", "k must be greater than 0");
  if (num_hashes == 0) raise_error("BatchNtHash", "num_hashes must be greater than 0");
}

BatchNtHash::~BatchNtHash() = default;

void BatchNtHash::add(const char* seq, size_t seq_len)
{
  seqs_.insert(seqs_.end(), seq, seq + seq_len);
  offsets_.push_back(seqs_.size());
}

void BatchNtHash::clear()
{
  seqs_.clear();
  offsets_.assign(1, 0);
  counts_.clear();
  first_.clear();
  hashes_.clear();
  pos_.clear();
  total_ = 0;
}

void BatchNtHash::run()
{
  const size_t n = size();
  counts_.assign(n, 0);
  first_.assign(n, 0);
  total_ = 0;
  size_t cap = 0;
  for (size_t r = 0; r < n; ++r) {
    const size_t len = (size_t)(offsets_[r + 1] - offsets_[r]);
    if (len >= k_) cap += len - k_ + 1;
  }
  hashes_.resize(cap * num_hashes_);
  pos_.resize(cap);
  if (n == 0 || cap == 0) return;
  nthip_reads rd = { seqs_.data(), offsets_.data(), (uint64_t)n, 0, 0 };
  nthip_out out = { hashes_.data(), (uint64_t)cap, counts_.data(), pos_.data(), nullptr, nullptr };
  uint64_t total = 0;
  int rc;
  if (nthip_multi* multi = device_multi("BatchNtHash"))
    rc = nthip_multi_kmer_hash(multi, &rd, (uint16_t)k_, (uint8_t)num_hashes_, &out, &total);
  else
    rc = nthip_kmer_hash(device_ctx("BatchNtHash"), &rd, (uint16_t)k_, (uint8_t)num_hashes_, &out, &total,
                         NTHIP_HOST_INPUT | NTHIP_HOST_OUTPUT);
  if (rc != NTHIP_OK) raise_error("BatchNtHash", std::string("GPU hashing failed: ") + nthip_last_error());
  total_ = total;
  uint64_t acc = 0;
  for (size_t r = 0; r < n; ++r) {
    first_[r] = acc;
    acc += counts_[r];
  }
  hashes_.resize((size_t)total * num_hashes_);
  pos_.resize((size_t)total);
}

} // namespace nthash
