// include/nthash/nthash.hpp -- nthash_amd's C++ host API.
//
// Drop-in for the public header of bcgsc/ntHash 2.4.0
// (reference: include/nthash/nthash.hpp): the same namespace, class names,
// constructor and method signatures, argument meaning, return values and error
// behaviour (misuse prints "[ntHash::<Class>] ERROR: ..." and exits with status
// 1, reference src/internal.hpp:16-22), so code written against the reference --
// including its own tests/tests.cpp and examples/*.cpp -- compiles against this
// header unchanged and links with libnthash.so (nthash_amd/lib).
//
// What is different is underneath.  NtHash::roll() and SeedNtHash::roll() do
// not roll a scalar state on the CPU: the first roll() hashes the WHOLE sequence
// on the MI355X through the C-ABI (include/nthash_hip.h: nthip_kmer_hash /
// nthip_seed_hash) and every roll() then steps through that device-computed
// stream.  There is no CPU fallback for this path: without a HIP device the
// first roll() reports the error and exits, like any other misuse.  The calls
// that cannot be batched -- roll_back(), the peek*() family and the Blind*
// classes, which hash ONE caller-chosen base per call -- evaluate the O(1)
// recurrence on the host (reference src/kmer.cpp:84-194, src/seed.cpp:177-425).
//
// For throughput use the batch C-ABI directly (one call per read set, device
// pointers in and out); this iterator API is the compatibility boundary.
#pragma once

#include <array>
#include <cstdint>
#include <cstring>
#include <deque>
#include <memory>
#include <string>
#include <sys/types.h>
#include <vector>

namespace nthash {

// reference: include/nthash/nthash.hpp:18
static const char* const NTHASH_FN_NAME = "ntHash_v2";

// reference: include/nthash/nthash.hpp:24-29
namespace typedefs {
using NUM_HASHES_TYPE = uint8_t;
using K_TYPE = uint16_t;
using SpacedSeedBlocks = std::vector<std::array<unsigned, 2>>;
using SpacedSeedMonomers = std::vector<unsigned>;
} // namespace typedefs

class NtHash;
class BlindNtHash;
class SeedNtHash;
class BlindSeedNtHash;

// Seed patterns -> lists of don't-care positions (reference nthash.hpp:59-60)
std::vector<std::vector<unsigned>>
parse_seeds(const std::vector<std::string>& seed_strings);

namespace detail {
struct KmerStream; // device-computed hash stream of one window of a sequence (opaque)
struct KmerAhead;  // the NEXT window's stream, being computed on the thread's helper while this one is walked
struct RollTables; // per k: what a byte adds when it enters / leaves a window (host recurrences)
struct SeedStream;
struct SeedAhead;  // ... and the same for a SeedNtHash (round 4)
struct SeedSet;    // parsed seeds: blocks, monomers, masks (host) + device tables
} // namespace detail

// ---------------------------------------------------------------------------
// Contiguous k-mer hashing (reference nthash.hpp:62-211, src/kmer.cpp:200-336)
// ---------------------------------------------------------------------------
class NtHash
{
public:
  NtHash(const char* seq,
         size_t seq_len,
         typedefs::NUM_HASHES_TYPE num_hashes,
         typedefs::K_TYPE k,
         size_t pos = 0);
  NtHash(const std::string& seq,
         typedefs::NUM_HASHES_TYPE num_hashes,
         typedefs::K_TYPE k,
         size_t pos = 0)
    : NtHash(seq.data(), seq.size(), num_hashes, k, pos)
  {
  }
  NtHash(const NtHash& obj);
  NtHash(NtHash&&) noexcept;
  ~NtHash();

  bool roll()
  {
    // The walk through a device-computed window, inline: when the stream's next entry IS the next position, that window
    // holds bases only (the stream holds nothing else: reference NtHash::roll, src/kmer.cpp:246-264, would roll into it)
    // and its hashes are the entry's.  Everything else -- the first call, a skip, the end of a window, an object whose
    // strand hashes somebody reads -- takes the general routine.
    const size_t i = cursor_ + 1;
    if (sp_ != nullptr && i < sn_ && sp_[i] == pos_ + 1 - sbegin_ && !strands_wanted_) {
      cursor_ = i;
      ++pos_;
      const uint64_t* h = sh_ + i * num_hashes_;
      uint64_t* out = hash_arr_.get();
      for (unsigned j = 0; j < num_hashes_; ++j) out[j] = h[j];
      strands_stale_ = true;
      return true;
    }
    return roll_general();
  }
  bool roll_back();
  bool peek();
  bool peek_back();
  bool peek(char char_in);
  bool peek_back(char char_in);

  const uint64_t* hashes() const { return hash_arr_.get(); }
  size_t get_pos() const { return pos_; }
  typedefs::NUM_HASHES_TYPE get_hash_num() const { return num_hashes_; }
  typedefs::K_TYPE get_k() const { return k_; }
  uint64_t get_forward_hash() const { if (strands_stale_) sync_strands(); return fwd_; }
  uint64_t get_reverse_hash() const { if (strands_stale_) sync_strands(); return rev_; }

private:
  const char* seq_;
  size_t len_;
  typedefs::NUM_HASHES_TYPE num_hashes_;
  typedefs::K_TYPE k_;
  size_t pos_;
  bool initialized_;
  // The strand hashes are not part of what the device sends back (hashes() is: 8 m bytes per k-mer instead of
  // 16 + 8 m, and two device passes fewer): a window taken from the device stream leaves them stale, the first call that
  // needs them -- these getters, roll_back(), peek*() -- hashes the current window's strands on the host (k steps), and
  // from then on roll() keeps them current with the O(1) recurrence.
  mutable uint64_t fwd_ = 0;
  mutable uint64_t rev_ = 0;
  mutable bool strands_stale_ = false;
  mutable bool strands_wanted_ = false;
  void sync_strands() const;
  std::unique_ptr<uint64_t[]> hash_arr_;
  const detail::RollTables* rt_ = nullptr;     // process-wide, immutable
  std::shared_ptr<detail::KmerStream> stream_; // shared by copies, immutable once built
  size_t cursor_ = 0;                          // last stream entry used (search hint)
  std::shared_ptr<detail::KmerAhead> ahead_;   // (not copied: a copy starts its own)
  // views into *stream_ for the inline walk of roll(): positions (relative to sbegin_), hashes, entries
  const uint32_t* sp_ = nullptr;
  const uint64_t* sh_ = nullptr;
  size_t sn_ = 0, sbegin_ = 0;

  bool init();
  bool roll_general();
  bool load_from_stream();
  void next_stream();
};

// ---------------------------------------------------------------------------
// Caller-fed k-mer hashing (reference nthash.hpp:213-311, src/kmer.cpp:338-393)
// ---------------------------------------------------------------------------
class BlindNtHash
{
public:
  BlindNtHash(const char* seq,
              typedefs::NUM_HASHES_TYPE num_hashes,
              typedefs::K_TYPE k,
              ssize_t pos = 0);
  BlindNtHash(const BlindNtHash& obj);
  BlindNtHash(BlindNtHash&&) = default;

  void roll(char char_in);
  void roll_back(char char_in);
  void peek(char char_in);
  void peek_back(char char_in);

  const uint64_t* hashes() const { return hash_arr_.get(); }
  ssize_t get_pos() const { return pos_; }
  typedefs::NUM_HASHES_TYPE get_hash_num() const { return num_hashes_; }
  typedefs::K_TYPE get_k() const { return (typedefs::K_TYPE)window_.size(); }
  uint64_t get_forward_hash() const { return fwd_; }
  uint64_t get_reverse_hash() const { return rev_; }

private:
  std::deque<char> window_;
  typedefs::NUM_HASHES_TYPE num_hashes_;
  ssize_t pos_;
  uint64_t fwd_ = 0;
  uint64_t rev_ = 0;
  std::unique_ptr<uint64_t[]> hash_arr_;
  const detail::RollTables* rt_ = nullptr;
};

// ---------------------------------------------------------------------------
// Spaced-seed hashing (reference nthash.hpp:313-521, src/seed.cpp:449-667)
// ---------------------------------------------------------------------------
class SeedNtHash
{
public:
  SeedNtHash(const char* seq,
             size_t seq_len,
             const std::vector<std::string>& seeds,
             typedefs::NUM_HASHES_TYPE num_hashes_per_seed,
             typedefs::K_TYPE k,
             size_t pos = 0);
  SeedNtHash(const std::string& seq,
             const std::vector<std::string>& seeds,
             typedefs::NUM_HASHES_TYPE num_hashes_per_seed,
             typedefs::K_TYPE k,
             size_t pos = 0)
    : SeedNtHash(seq.data(), seq.size(), seeds, num_hashes_per_seed, k, pos)
  {
  }
  SeedNtHash(const char* seq,
             size_t seq_len,
             const std::vector<std::vector<unsigned>>& seeds,
             typedefs::NUM_HASHES_TYPE num_hashes_per_seed,
             typedefs::K_TYPE k,
             size_t pos = 0);
  SeedNtHash(const std::string& seq,
             const std::vector<std::vector<unsigned>>& seeds,
             typedefs::NUM_HASHES_TYPE num_hashes_per_seed,
             typedefs::K_TYPE k,
             size_t pos = 0)
    : SeedNtHash(seq.data(), seq.size(), seeds, num_hashes_per_seed, k, pos)
  {
  }
  SeedNtHash(const SeedNtHash& obj);
  SeedNtHash(SeedNtHash&&) noexcept;
  ~SeedNtHash();

  bool roll()
  {
    // the walk through a device-computed window, inline (as NtHash::roll): when the stream's next entry IS the next position
    // the reference's roll() goes there too (src/seed.cpp:518-544: the incoming character is a base -- otherwise it would
    // jump k positions, which is the next position only for k = 1) and the entry holds its hashes
    const size_t i = cursor_ + 1;
    if (sp_ != nullptr && i < sn_ && sp_[i] == pos_ + 1 - sbegin_ && k_ > 1) {
      cursor_ = i;
      ++pos_;
      const unsigned n = get_hash_num();
      const uint64_t* h = sh_ + i * n;
      uint64_t* out = hash_arr_.get();
      for (unsigned j = 0; j < n; ++j) out[j] = h[j];
      strands_stale_ = true;
      return true;
    }
    return roll_general();
  }
  bool roll_back();
  bool peek();
  bool peek_back();
  bool peek(char char_in);
  bool peek_back(char char_in);

  const uint64_t* hashes() const { return hash_arr_.get(); }
  size_t get_pos() const { return pos_; }
  unsigned get_hash_num() const { return num_hashes_per_seed_ * n_seeds_; }
  typedefs::NUM_HASHES_TYPE get_hash_num_per_seed() const { return num_hashes_per_seed_; }
  typedefs::K_TYPE get_k() const { return k_; }
  uint64_t* get_forward_hash() const { if (strands_stale_) sync_strands(); return fwd_.get(); }
  uint64_t* get_reverse_hash() const { if (strands_stale_) sync_strands(); return rev_.get(); }

private:
  const char* seq_;
  size_t len_;
  typedefs::NUM_HASHES_TYPE num_hashes_per_seed_;
  typedefs::K_TYPE k_;
  size_t pos_;
  size_t pos0_; // where the device stream starts (constructor's pos)
  // (as in NtHash: the device stream carries hashes() only; the per-seed strand hashes of the current window are
  //  computed on the host when a getter asks for them)
  mutable bool strands_stale_ = false;
  void sync_strands() const;
  bool initialized_;
  unsigned n_seeds_;
  std::shared_ptr<detail::SeedSet> seeds_;
  std::unique_ptr<uint64_t[]> fwd_;
  std::unique_ptr<uint64_t[]> rev_;
  std::unique_ptr<uint64_t[]> hash_arr_;
  std::shared_ptr<detail::SeedStream> stream_;
  std::shared_ptr<detail::SeedAhead> ahead_;   // the window after stream_, on the thread's helper (not copied)
  size_t cursor_ = 0;
  // views into *stream_ for the inline walk of roll(): positions (relative to sbegin_), hashes, entries
  const uint32_t* sp_ = nullptr;
  const uint64_t* sh_ = nullptr;
  size_t sn_ = 0, sbegin_ = 0;

  bool roll_general();
  bool init(bool from_roll);
  void set_window(const char* win, bool try_stream);
  void hash_backward(bool commit);
};

// ---------------------------------------------------------------------------
// Caller-fed spaced-seed hashing (reference nthash.hpp:523-646, src/seed.cpp:669-737)
// ---------------------------------------------------------------------------
class BlindSeedNtHash
{
public:
  BlindSeedNtHash(const char* seq,
                  const std::vector<std::string>& seeds,
                  typedefs::NUM_HASHES_TYPE num_hashes_per_seed,
                  typedefs::K_TYPE k,
                  ssize_t pos = 0);
  BlindSeedNtHash(const BlindSeedNtHash& seed_nthash);
  BlindSeedNtHash(BlindSeedNtHash&&) = default;

  void roll(char char_in);
  void roll_back(char char_in);

  const uint64_t* hashes() const { return hash_arr_.get(); }
  ssize_t get_pos() const { return pos_; }
  unsigned get_hash_num() const { return num_hashes_per_seed_ * n_seeds_; }
  typedefs::NUM_HASHES_TYPE get_hash_num_per_seed() const { return num_hashes_per_seed_; }
  typedefs::K_TYPE get_k() const { return k_; }
  uint64_t* get_forward_hash() const { return fwd_.get(); }
  uint64_t* get_reverse_hash() const { return rev_.get(); }

private:
  std::deque<char> window_;
  typedefs::NUM_HASHES_TYPE num_hashes_per_seed_;
  typedefs::K_TYPE k_;
  ssize_t pos_;
  unsigned n_seeds_;
  std::shared_ptr<detail::SeedSet> seeds_;
  std::unique_ptr<uint64_t[]> fwd_;
  std::unique_ptr<uint64_t[]> rev_;
  std::unique_ptr<uint64_t[]> hash_arr_;

  void rehash();
};

// ---------------------------------------------------------------------------
// BatchNtHash -- an ADDITION to the reference's API (nothing like it exists there): the loop every caller of the
// library writes,
//     for (read : reads) { NtHash h(read, m, k); while (h.roll()) use(h.get_pos(), h.hashes()); }
// (examples/benchmark.cpp:34-39 of the reference), as ONE device call for all the reads added so far.  The k-mers of
// read i come back in roll() order: count(i) of them, hashes(i)[j * num_hashes + h], positions(i)[j]; a read
// shorter than k simply has none (where the iterator's constructor would exit).  Use it where throughput matters:
// one object per read costs a device round trip each (or, for short reads, runs on the host).
// ---------------------------------------------------------------------------
class BatchNtHash
{
public:
  BatchNtHash(typedefs::NUM_HASHES_TYPE num_hashes, typedefs::K_TYPE k);
  ~BatchNtHash();
  BatchNtHash(const BatchNtHash&) = delete;
  BatchNtHash& operator=(const BatchNtHash&) = delete;

  void add(const char* seq, size_t seq_len); // the bytes are copied
  void add(const std::string& seq) { add(seq.data(), seq.size()); }
  size_t size() const { return offsets_.size() - 1; }
  void run();   // hashes every read added since the last clear(); results stay valid until the next add()/clear()
  void clear();

  size_t count(size_t read) const { return (size_t)counts_[read]; }
  const uint64_t* hashes(size_t read) const { return hashes_.data() + first_[read] * num_hashes_; }
  const uint32_t* positions(size_t read) const { return pos_.data() + first_[read]; }
  uint64_t total() const { return total_; }
  typedefs::NUM_HASHES_TYPE get_hash_num() const { return num_hashes_; }
  typedefs::K_TYPE get_k() const { return k_; }

private:
  typedefs::NUM_HASHES_TYPE num_hashes_;
  typedefs::K_TYPE k_;
  std::vector<char> seqs_;
  std::vector<uint64_t> offsets_; // size() + 1
  std::vector<uint64_t> counts_, first_, hashes_;
  std::vector<uint32_t> pos_;
  uint64_t total_ = 0;
};

} // namespace nthash
