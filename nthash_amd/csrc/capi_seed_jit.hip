// capi_seed_jit.hip -- the kernel specialisation cache: seed_ps_kernel.hpp compiled at run time for ONE
// seed set and read shape (hiprtc), kept per seed set in memory and per source text on disk.
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
//
// hiprtc is looked up when the first kernel is asked for (dlopen: the library itself does not depend on it); without it,
// or when a compile fails, the caller's precompiled kernels hash the batch -- same results, the specialised code is only
// faster.  NTHIP_SEED_JIT=0: never; =1: for every batch the specialised kernel can take, whatever its size (tests).
#include "capi_internal.hpp"

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <fstream>
#include <mutex>
#include <atomic>
#include <thread>
#include <sstream>

using namespace ntamd;
using namespace ntamd::host;

namespace {

static const char PSJ_SRC[] =
#include "seed_psj_kernel.inc"
    ;

struct Rtc {
  void* lib = nullptr;
  bool tried = false;
  int (*create)(void**, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
  int (*compile)(void*, int, const char* const*) = nullptr;
  int (*log_size)(void*, size_t*) = nullptr;
  int (*log)(void*, char*) = nullptr;
  int (*code_size)(void*, size_t*) = nullptr;
  int (*code)(void*, char*) = nullptr;
  int (*destroy)(void**) = nullptr;
};
Rtc g_rtc;
std::mutex g_rtc_mu;

bool rtc_ready()
{
  std::lock_guard<std::mutex> lk(g_rtc_mu);
  if (g_rtc.tried) return g_rtc.lib != nullptr;
  g_rtc.tried = true;
  for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
    g_rtc.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (g_rtc.lib) break;
  }
  if (!g_rtc.lib) return false;
  auto sym = [&](const char* n) { return dlsym(g_rtc.lib, n); };
  g_rtc.create = (decltype(g_rtc.create))sym("hiprtcCreateProgram");
  g_rtc.compile = (decltype(g_rtc.compile))sym("hiprtcCompileProgram");
  g_rtc.log_size = (decltype(g_rtc.log_size))sym("hiprtcGetProgramLogSize");
  g_rtc.log = (decltype(g_rtc.log))sym("hiprtcGetProgramLog");
  g_rtc.code_size = (decltype(g_rtc.code_size))sym("hiprtcGetCodeSize");
  g_rtc.code = (decltype(g_rtc.code))sym("hiprtcGetCode");
  g_rtc.destroy = (decltype(g_rtc.destroy))sym("hiprtcDestroyProgram");
  if (!g_rtc.create || !g_rtc.compile || !g_rtc.code_size || !g_rtc.code || !g_rtc.destroy) {
    dlclose(g_rtc.lib);
    g_rtc.lib = nullptr;
  }
  return g_rtc.lib != nullptr;
}

uint64_t fnv1a(const std::string& s)
{
  uint64_t h = 1469598103934665603ull;
  for (unsigned char ch : s) {
    h ^= ch;
    h *= 1099511628211ull;
  }
  return h;
}

std::string cache_dir()
{
  const char* d = getenv("NTHIP_JIT_CACHE");
  if (d && !*d) return ""; // (empty: no disk cache)
  std::string dir;
  if (d) dir = d;
  else if (const char* x = getenv("XDG_CACHE_HOME")) dir = std::string(x) + "/nthash_amd";
  else if (const char* h = getenv("HOME")) {
    dir = std::string(h) + "/.cache";
    (void)mkdir(dir.c_str(), 0755); // (a fresh account has no ~/.cache yet)
    dir += "/nthash_amd";
  } else return "";
  (void)mkdir(dir.c_str(), 0755);
  return dir;
}

std::string cache_path(const std::string& src)
{
  char tag[64];
  snprintf(tag, sizeof tag, "psj_%016llx_%zu.hsaco", (unsigned long long)fnv1a(src), src.size());
  const std::string dir = cache_dir();
  return dir.empty() ? "" : dir + "/" + tag;
}

// the code object of a source text that was compiled before (the disk cache): milliseconds, on the caller's thread
bool cached_code(const std::string& src, std::vector<char>* code)
{
  const std::string path = cache_path(src);
  if (path.empty()) return false;
  std::ifstream in(path, std::ios::binary);
  if (!in) return false;
  code->assign(std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>());
  return code->size() > 64;
}

// source text -> code object for gfx950 (from the disk cache when the same text was compiled before)
bool compile_source(const std::string& src, std::vector<char>* code, std::string* why)
{
  if (cached_code(src, code)) return true;
  const std::string path = cache_path(src);
  if (!rtc_ready()) {
    *why = "libhiprtc.so not found";
    return false;
  }
  void* prog = nullptr;
  if (g_rtc.create(&prog, src.c_str(), "seed_psj.hip", 0, nullptr, nullptr) != 0) {
    *why = "hiprtcCreateProgram failed";
    return false;
  }
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed"};
  const int rc = g_rtc.compile(prog, 4, opts);
  if (rc != 0) {
    size_t n = 0;
    if (g_rtc.log_size && g_rtc.log && g_rtc.log_size(prog, &n) == 0 && n > 1) {
      std::string log(n, '\0');
      g_rtc.log(prog, &log[0]);
      *why = "hiprtc: " + log.substr(0, 2000);
    } else *why = "hiprtcCompileProgram failed";
    g_rtc.destroy(&prog);
    return false;
  }
  size_t n = 0;
  if (g_rtc.code_size(prog, &n) != 0 || n == 0) {
    *why = "hiprtcGetCodeSize failed";
    g_rtc.destroy(&prog);
    return false;
  }
  code->resize(n);
  g_rtc.code(prog, code->data());
  g_rtc.destroy(&prog);
  if (!path.empty()) { // (written under another name first: a reader never sees half a file)
    const std::string tmp = path + "." + std::to_string((long)getpid());
    std::ofstream out(tmp, std::ios::binary);
    if (out) {
      out.write(code->data(), (std::streamsize)code->size());
      out.close();
      if (rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());
    }
  }
  return true;
}

} // namespace

std::string ntamd::host::seed_psj_source(const SeedJitShape& g)
{
  std::ostringstream o;
  o << "#define PSJ_LEN " << g.len << "u\n#define PSJ_K " << g.k << "u\n#define PSJ_NWIN " << g.nwin << "u\n#define PSJ_M2 " << g.m2
    << "u\n#define PSJ_NSEEDS " << g.n_seeds << "u\n#define PSJ_W " << g.W << "u\n#define PSJ_NB_LOG " << g.nb_log
    << "\n#define PSJ_LPR_LOG " << g.lpr_log << "\n#define PSJ_NARR " << g.n_arrays << "\n#define PSJ_SEGS_B " << g.segs_b
    << "u\n#define PSJ_NT " << g.term_arr.size() << "u\n#define PSJ_WAVES " << g.waves << "u\n#define PSJ_KEEP " << (g.W <= 8 ? 1 : 0)
    << "\n";
  auto arr = [&](const char* type, const char* name, const std::vector<uint64_t>& v, const char* suffix) {
    o << "__device__ constexpr " << type << " " << name << "[] = {";
    for (size_t i = 0; i < v.size(); ++i) o << (i ? ", " : "") << v[i] << suffix;
    o << "};\n";
  };
  arr("unsigned int", "PSJ_T_ARR", std::vector<uint64_t>(g.term_arr.begin(), g.term_arr.end()), "u");
  arr("unsigned int", "PSJ_T_E", std::vector<uint64_t>(g.term_e.begin(), g.term_e.end()), "u");
  arr("unsigned int", "PSJ_SEED_FIRST", std::vector<uint64_t>(g.seed_first.begin(), g.seed_first.end()), "u");
  std::vector<uint64_t> mult;
  for (uint32_t i = 0; i < g.m2; ++i) mult.push_back(multiplier(g.k, i));
  arr("unsigned long long", "PSJ_MULT", mult, "ull");
  o << PSJ_SRC;
  return o.str();
}

// The specialised kernel of a shape: from the process's registry, the disk, or the compiler.  nullptr (and *why): not
// available (yet).  A code object is loaded ONCE per process and device and stays loaded: seed sets of the same strings share
// it (loading and unloading the same image again and again -- a seed set per call, several contexts -- ended in memory faults
// of the launched kernels on ROCm 7.2; the registry is bounded by the distinct seed sets and read shapes a process meets).
// wait == false: a compile that has to be made runs on a thread of its own and THIS call returns nullptr -- the caller's
// precompiled kernels hash the batch; a later call finds the code object and loads it.  A second or two of compiler are
// then never on anybody's clock.  wait == true (NTHIP_SEED_JIT=1): compile here and now.
namespace {
struct Loaded {
  enum State { COMPILING, COMPILED, READY, FAILED } state = COMPILING;
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
  std::vector<char> image; // (the runtime may load from the image lazily: it lives as long as the module)
  std::string why;
};
// (never destroyed: a compile thread may outlive main())
std::map<std::string, Loaded>& registry() { static auto* r = new std::map<std::string, Loaded>(); return *r; }
std::mutex& registry_mu() { static auto* m = new std::mutex(); return *m; }

// A process that ends while a compile thread is inside the compiler would pull the compiler's own statics from under it:
// exit() waits for the compiles in flight (a second or two; bounded).  Registered AFTER libhiprtc.so is loaded, so it runs
// BEFORE that library's destructors.
std::atomic<int> g_compiles_in_flight{0};
void wait_for_compiles()
{
  for (int i = 0; i < 6000 && g_compiles_in_flight.load() > 0; ++i) std::this_thread::sleep_for(std::chrono::milliseconds(5));
}
void arm_exit_wait()
{
  static std::once_flag once;
  std::call_once(once, [] {
    (void)rtc_ready();
    atexit(wait_for_compiles);
  });
}

// COMPILED -> READY / FAILED, on the caller's thread (its device is current)
void load_module(Loaded& L)
{
  if (hipModuleLoadData(&L.mod, L.image.data()) != hipSuccess || hipModuleGetFunction(&L.fn, L.mod, "psj") != hipSuccess) {
    (void)hipGetLastError();
    L.why = "hipModuleLoadData / hipModuleGetFunction failed";
    L.state = Loaded::FAILED;
    return;
  }
  int scratch = 0; // (a kernel that spills is slower than the precompiled one it replaces)
  if (hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, L.fn) == hipSuccess && scratch > 0) {
    L.why = "the specialised kernel spills registers";
    L.fn = nullptr; // (stays loaded, never launched)
    L.state = Loaded::FAILED;
    return;
  }
  L.state = Loaded::READY;
}
} // namespace
void* ntamd::host::seed_psj_get(nthip_ctx* c, const nthip_seeds* sd, const SeedJitShape& g, bool wait, std::string* why)
{
  (void)sd;
  const std::string src = seed_psj_source(g);
  char keybuf[48];
  snprintf(keybuf, sizeof keybuf, "%016llx_%zu_%d", (unsigned long long)fnv1a(src), src.size(), c->device);
  const std::string key = keybuf;
  std::unique_lock<std::mutex> lk(registry_mu());
  auto& reg = registry();
  auto it = reg.find(key);
  if (it == reg.end()) {
    Loaded& L = reg[key]; // (COMPILING)
    std::vector<char> known;
    if (wait || cached_code(src, &known)) { // (a code object on the disk is read here and now, whatever `wait` says)
      lk.unlock();
      std::vector<char> image;
      std::string w;
      bool ok = !known.empty();
      if (ok) image.swap(known);
      else ok = compile_source(src, &image, &w);
      lk.lock();
      Loaded& M = reg[key];
      if (ok) {
        M.image.swap(image);
        M.state = Loaded::COMPILED;
      } else {
        M.why = w;
        M.state = Loaded::FAILED;
      }
    } else {
      (void)L;
      arm_exit_wait();
      g_compiles_in_flight.fetch_add(1);
      std::thread([src, key]() {
        std::vector<char> image;
        std::string w;
        const bool ok = compile_source(src, &image, &w);
        {
          std::lock_guard<std::mutex> g2(registry_mu());
          Loaded& M = registry()[key];
          if (ok) {
            M.image.swap(image);
            M.state = Loaded::COMPILED;
          } else {
            M.why = w;
            M.state = Loaded::FAILED;
          }
        }
        g_compiles_in_flight.fetch_sub(1);
      }).detach();
      *why = "being compiled";
      return nullptr;
    }
    it = reg.find(key);
  }
  Loaded& L = it->second;
  if (L.state == Loaded::COMPILING) {
    if (!wait) {
      *why = "being compiled";
      return nullptr;
    }
    while (L.state == Loaded::COMPILING) { // (another thread's compile: wait for it)
      lk.unlock();
      std::this_thread::sleep_for(std::chrono::milliseconds(5));
      lk.lock();
    }
  }
  if (L.state == Loaded::COMPILED) load_module(L);
  if (L.state == Loaded::READY) return (void*)L.fn;
  *why = L.why.empty() ? "no specialised kernel" : L.why;
  return nullptr;
}

void ntamd::host::seed_jit_release(const nthip_seeds* sd) { (void)sd; }

// The source text that is compiled for a seed set on reads of `len` bases (malloc'ed: free() it) -- no device needed: what
// tests/test_seed_jit_source.py compiles with hipcc and what a user may want to look at.
extern "C" int nthip_seed_jit_source(const char* const* seeds, uint32_t n_seeds, uint16_t k16, uint32_t len, uint8_t m2, char** out)
{
  if (!seeds || !out || n_seeds == 0) return fail(NTHIP_ERR_ARG, "seeds/out is NULL");
  *out = nullptr;
  nthip_ctx tmp; // (never touches a device: the tuning knobs of the environment and the LDS of gfx950)
  tmp.lds_max = 160 * 1024;
  load_tuning(tmp.tune);
  nthip_seeds sd;
  sd.k = k16;
  sd.n_seeds = n_seeds;
  for (uint32_t s = 0; s < n_seeds; ++s) {
    if (!seeds[s] || strlen(seeds[s]) != k16) return fail(NTHIP_ERR_ARG, "seed %u is not k = %u characters", s, (unsigned)k16);
    std::vector<uint8_t> care(k16);
    for (uint32_t p = 0; p < k16; ++p) care[p] = seeds[s][p] == '1'; // (src/seed.cpp:25-50: any other character is a don't-care)
    sd.h_care.push_back(care);
  }
  SeedJitShape g;
  if (!seed_jit_shape(&tmp, &sd, len, m2, &g)) return fail(NTHIP_ERR_UNSUPPORTED, "no specialised kernel for this seed set and read length");
  const std::string s = seed_psj_source(g);
  *out = (char*)malloc(s.size() + 1);
  if (!*out) return fail(NTHIP_ERR_HIP, "out of memory");
  memcpy(*out, s.c_str(), s.size() + 1);
  return NTHIP_OK;
}
