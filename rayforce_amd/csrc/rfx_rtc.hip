// rfx_rtc.hip -- kernels compiled at run time for ONE plan (hiprtc), cached by plan signature.
//
// The prebuilt kernels decode their plan (which column feeds which aggregate, of which kind, under which predicates) from
// descriptors at run time and must keep per-group state where a run-time index can reach it (LDS).  A kernel generated for one
// plan has every descriptor as a constant and can keep per-(aggregate, group) state in registers: rfx_group_few_rtc.hpp.
// libhiprtc.so is loaded on first use (dlopen: nothing links against it, a box without it simply keeps the prebuilt kernels).  The
// generated text #includes the library's own kernel headers, which are EMBEDDED in librfx.so (build/rfx_rtc_headers.inc, made from
// the sources by the Makefile) and handed to hiprtcCreateProgram by name: the library needs no source tree beside it.  A plan whose
// kernel cannot be built (no compiler, a compile error) runs the prebuilt kernel: RFX_ESTATE from here means exactly that.
// Code objects are kept ON DISK, keyed by a hash of the generated text + the embedded headers + the compiler's version + the
// options: <dir of librfx.so>/rtc_cache (RFX_RTC_CACHE=<dir> moves it, RFX_RTC_CACHE=0 turns it off).  A plan found there is loaded at
// first sight (milliseconds); only a plan that is not waits until it has come back over enough rows and then pays the compilation
// (seconds; RFX_TRACE=1 says so).  rfx_hip_rtc_prewarm_* compile into the cache without a device (the build step runs them for the
// BASELINE plans, the directory travels with the library).
#include <dlfcn.h>
#include <stdlib.h>
#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include "rfx_group_common.hpp"
#include "rfx_where_once_kernel.hpp"
#include "build/rfx_rtc_headers.inc" // RTC_NHDR, RTC_HDR_NAMES[], RTC_HDR_TEXTS[]: the kernel headers as text

typedef struct _hiprtcProgram *rtcProgram;
static struct {
    int state; // 0 untried, 1 ready, -1 unavailable
    int (*create)(rtcProgram *, const char *, const char *, int, const char **, const char **);
    int (*compile)(rtcProgram, int, const char **);
    int (*log_size)(rtcProgram, size_t *);
    int (*log)(rtcProgram, char *);
    int (*code_size)(rtcProgram, size_t *);
    int (*code)(rtcProgram, char *);
    int (*destroy)(rtcProgram *);
    int (*version)(int *, int *);
    std::string cache_dir; // "": no disk cache
    u64 env_hash[2];       // embedded headers + compiler version + options
    int cache_files;
} R;
static std::map<std::string, hipFunction_t> *g_cache; // signature -> kernel (NULL: failed once, never retried)
static long long g_launches, g_compiles, g_disk_loads, g_disk_writes;
extern "C" void rfx_hip_rtc_stats(int64_t *launches, int64_t *compiles) {
    if (launches) *launches = g_launches;
    if (compiles) *compiles = g_compiles;
}
extern "C" void rfx_hip_rtc_cache_stats(int64_t *loaded_from_disk, int64_t *written_to_disk) {
    if (loaded_from_disk) *loaded_from_disk = g_disk_loads;
    if (written_to_disk) *written_to_disk = g_disk_writes;
}
static const char *const RTC_OPTS[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-fast-math", "-ffp-contract=off"};
static void fnv2(u64 h[2], const void *p, size_t n) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; i++) {
        h[0] = (h[0] ^ b[i]) * 0x100000001B3ULL;
        h[1] = (h[1] ^ (b[i] + 0x9Eu)) * 0x00000100000001B3ULL + 0x9E3779B97F4A7C15ULL;
    }
}

static bool trace() { return getenv("RFX_TRACE") != NULL; }
static bool rtc_ready_once();
static bool rtc_ready() {
    static std::mutex once;
    std::lock_guard<std::mutex> hold(once);
    return rtc_ready_once();
}
static bool rtc_ready_once() {
    if (R.state) return R.state > 0;
    R.state = -1;
    if (getenv("RFX_NO_RTC")) return false;
    void *h = dlopen("libhiprtc.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/libhiprtc.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        if (trace()) fprintf(stderr, "[rfx] rtc: libhiprtc.so not loadable (%s)\n", dlerror());
        return false;
    }
    *(void **)&R.create = dlsym(h, "hiprtcCreateProgram");
    *(void **)&R.compile = dlsym(h, "hiprtcCompileProgram");
    *(void **)&R.log_size = dlsym(h, "hiprtcGetProgramLogSize");
    *(void **)&R.log = dlsym(h, "hiprtcGetProgramLog");
    *(void **)&R.code_size = dlsym(h, "hiprtcGetCodeSize");
    *(void **)&R.code = dlsym(h, "hiprtcGetCode");
    *(void **)&R.destroy = dlsym(h, "hiprtcDestroyProgram");
    *(void **)&R.version = dlsym(h, "hiprtcVersion");
    if (!R.create || !R.compile || !R.log_size || !R.log || !R.code_size || !R.code || !R.destroy) return false;
    // what a cached code object depends on besides its own text
    R.env_hash[0] = 0xCBF29CE484222325ULL;
    R.env_hash[1] = 0x84222325CBF29CE4ULL;
    int vmaj = 0, vmin = 0;
    if (R.version) R.version(&vmaj, &vmin);
    fnv2(R.env_hash, &vmaj, sizeof(vmaj));
    fnv2(R.env_hash, &vmin, sizeof(vmin));
    for (size_t i = 0; i < sizeof(RTC_OPTS) / sizeof(RTC_OPTS[0]); i++) fnv2(R.env_hash, RTC_OPTS[i], strlen(RTC_OPTS[i]) + 1);
    for (int i = 0; i < RTC_NHDR; i++) fnv2(R.env_hash, RTC_HDR_TEXTS[i], strlen(RTC_HDR_TEXTS[i]) + 1);
    // the cache directory: RFX_RTC_CACHE, else rtc_cache beside the library, else a per-user one under /tmp
    const char *env = getenv("RFX_RTC_CACHE");
    R.cache_dir.clear();
    if (!env || (strcmp(env, "0") != 0 && env[0])) {
        std::string cand[3];
        int nc = 0;
        if (env) cand[nc++] = env;
        else {
            Dl_info di;
            if (dladdr((const void *)&rtc_ready, &di) && di.dli_fname) {
                std::string so(di.dli_fname);
                const size_t cut = so.find_last_of('/');
                cand[nc++] = (cut == std::string::npos ? std::string(".") : so.substr(0, cut)) + "/rtc_cache";
            }
            // a per-user directory: $XDG_CACHE_HOME/rfx_rtc, ~/.cache/rfx_rtc, and only then /tmp (a name anybody can guess: see the check below)
            const char *xdg = getenv("XDG_CACHE_HOME"), *home = getenv("HOME");
            if (xdg && xdg[0] == '/') cand[nc++] = std::string(xdg) + "/rfx_rtc";
            else if (home && home[0] == '/') cand[nc++] = std::string(home) + "/.cache/rfx_rtc";
            char tmp[64];
            snprintf(tmp, sizeof(tmp), "/tmp/rfx_rtc_cache_%u", (unsigned)getuid());
            cand[nc++] = tmp;
        }
        for (int i = 0; i < nc && R.cache_dir.empty(); i++) {
            (void)mkdir(cand[i].c_str(), 0700);
            // code objects found here are LOADED AND RUN: the directory must be a real directory of ours that nobody else can write into
            // (another local user could pre-create a guessable /tmp name and plant <hash>.co files); anything else: no disk cache
            struct stat st;
            if (lstat(cand[i].c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH))) continue;
            if (access(cand[i].c_str(), W_OK | X_OK) == 0) R.cache_dir = cand[i];
        }
        if (!R.cache_dir.empty()) {
            R.cache_files = 0;
            if (DIR *d = opendir(R.cache_dir.c_str())) {
                while (readdir(d)) R.cache_files++;
                closedir(d);
            }
        }
    }
    if (trace()) fprintf(stderr, "[rfx] rtc: ready, code objects cached in %s\n", R.cache_dir.empty() ? "(memory only)" : R.cache_dir.c_str());
    R.state = 1;
    return true;
}

// <cache dir>/<hash of the text and of everything else the code depends on>.co
static std::string cache_path(const std::string &src) {
    if (R.cache_dir.empty()) return std::string();
    u64 h[2] = {R.env_hash[0], R.env_hash[1]};
    fnv2(h, src.data(), src.size());
    char name[48];
    snprintf(name, sizeof(name), "/%016llx%016llx.co", h[0], h[1]);
    return R.cache_dir + name;
}
static bool read_code(const std::string &path, std::string &code) {
    FILE *f = path.empty() ? NULL : fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t got;
    code.clear();
    while ((got = fread(buf, 1, sizeof(buf), f)) > 0) code.append(buf, got);
    fclose(f);
    return code.size() > 64 && memcmp(code.data(), "\177ELF", 4) == 0;
}
static void write_code(const std::string &path, const std::string &code) {
    if (path.empty() || R.cache_files > 4096) return; // (a bounded directory: past that, plans compile per process as before)
    char tmp[32];
    snprintf(tmp, sizeof(tmp), ".%d.tmp", (int)getpid());
    const std::string t = path + tmp;
    FILE *f = fopen(t.c_str(), "wb");
    if (!f) return;
    const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
    if (fclose(f) != 0 || !ok || rename(t.c_str(), path.c_str()) != 0) {
        (void)unlink(t.c_str());
        return;
    }
    R.cache_files++;
    g_disk_writes++;
}

// The plan's descriptor part as a braced initialiser of `Plan` (field order of rfx_common.hpp), the key columns as macros: the text is
// both what the kernel is compiled from and -- with the sizes -- its cache key.  Run-time values stay out of it: column pointers,
// row counts, the predicates' atoms (only whether an f64 atom is NaN, which picks the comparison's code), kmin / multipliers.
static void plan_text(const Plan &P, const GroupArgs *G, std::string &o) {
    char b[512];
#define ADD(...) do { snprintf(b, sizeof(b), __VA_ARGS__); o += b; } while (0)
    ADD("#define RTC_PLAN { %d, %d, %d, %d, {}, { ", P.ncols, P.npred, P.nagg, P.logic);
    for (int i = 0; i < P.npred; i++) {
        const bool cnan = P.preds[i].rhs_col < 0 && P.preds[i].dom_f64 && (P.preds[i].rhs_bits & 0x7FF0000000000000ULL) == 0x7FF0000000000000ULL && (P.preds[i].rhs_bits & 0x000FFFFFFFFFFFFFULL) != 0;
        ADD("{ %d, %d, %d, %d, %d, %d, %s, %d, %d }, ", P.preds[i].col, P.preds[i].rhs_col, P.preds[i].op, P.preds[i].dom_f64, P.preds[i].lhs_cvt, P.preds[i].rhs_cvt,
            cnan ? "0x7FF8000000000000ULL" : "0ULL", P.preds[i].more, P.preds[i].tree);
    }
    o += "}, { ";
    for (int a = 0; a < P.nagg; a++) ADD("{ %d, %d, %d, %d }, ", P.aggs[a].col, P.aggs[a].f64, P.aggs[a].kind, P.aggs[a].skipnull);
    ADD("}, 0, 0, %d, 0, { ", P.nx);
    for (int x = 0; x < P.nx; x++) {
        ADD("{ %d, %d, { ", P.xs[x].nops, P.xs[x].out_f64);
        for (int j = 0; j < P.xs[x].nops; j++) {
            const PlanXNode &n = P.xs[x].ops[j];
            ADD("{ %d, %d, %d, %d, %d, %d, %d, %d, 0x%llxULL, 0x%llxULL }, ", n.op, n.o_f64, n.l_kind, n.r_kind, n.l_idx, n.r_idx, n.l_f64, n.r_f64, (unsigned long long)n.l_atom,
                (unsigned long long)n.r_atom);
        }
        o += "} }, ";
    }
    o += "} }\n";
    if (G) {
        ADD("#define FEW_KEY_IDX %d\n#define FEW_NKEYS %d\n#define FEW_KIDX { ", G->key_idx, G->nkeys);
        for (int i = 0; i < G->nkeys; i++) ADD("%d, ", G->kidx[i]);
        o += "}\n";
    }
#undef ADD
}

static bool compile(const std::string &src, std::string &code) {
    rtcProgram prog = NULL;
    if (R.create(&prog, src.c_str(), "rfx_plan_kernel.hip", RTC_NHDR, (const char **)RTC_HDR_TEXTS, (const char **)RTC_HDR_NAMES) != 0) return false;
    const int rc = R.compile(prog, (int)(sizeof(RTC_OPTS) / sizeof(RTC_OPTS[0])), (const char **)RTC_OPTS);
    g_compiles++;
    if (rc != 0) {
        size_t ls = 0;
        R.log_size(prog, &ls);
        std::string log(ls + 1, 0);
        if (ls) R.log(prog, &log[0]);
        if (trace()) fprintf(stderr, "[rfx] rtc: compile failed (%d): %.2000s\n", rc, log.c_str());
        R.destroy(&prog);
        return false;
    }
    size_t cs = 0;
    R.code_size(prog, &cs);
    code.assign(cs, 0);
    R.code(prog, &code[0]);
    R.destroy(&prog);
    return cs > 0;
}
static hipFunction_t load(const std::string &code, const char *name) {
    hipModule_t mod;
    hipFunction_t fn = NULL;
    if (hipModuleLoadData(&mod, code.data()) != hipSuccess || hipModuleGetFunction(&fn, mod, name) != hipSuccess) {
        if (trace()) fprintf(stderr, "[rfx] rtc: module load failed\n");
        return NULL;
    }
    return fn; // (the module lives as long as the process: a handful of plans)
}

// the kernel of a generated text: from the cache, or compiled now if the plan has come back often enough over enough rows
// (a loaded kernel belongs to ONE device: the key carries the device ordinal; one lock around the maps -- a host that drives several
//  devices plans from one thread per device, rfx_exec.c)
static std::mutex g_rtc_lock;
static hipFunction_t plan_kernel_locked(const std::string &sig, const std::string &src, const char *name, i64 nrows, const char *what);
static hipFunction_t plan_kernel(int device, const std::string &sig0, const std::string &src, const char *name, i64 nrows, const char *what) {
    std::lock_guard<std::mutex> hold(g_rtc_lock);
    char dv[24];
    snprintf(dv, sizeof(dv), "//dev%d\n", device);
    return plan_kernel_locked(std::string(dv) + sig0, src, name, nrows, what);
}
static hipFunction_t plan_kernel_locked(const std::string &sig, const std::string &src, const char *name, i64 nrows, const char *what) {
    if (!g_cache) g_cache = new std::map<std::string, hipFunction_t>();
    auto it = g_cache->find(sig);
    if (it != g_cache->end()) return it->second;
    static std::map<std::string, int> *seen;
    if (!seen) seen = new std::map<std::string, int>();
    const int times = ++(*seen)[sig];
    const std::string path = cache_path(src);
    if (times == 1) { // development: RFX_RTC_DUMP=<dir> keeps every plan's generated text (compile it offline with hipcc -S to read its ISA / registers)
        if (const char *dump = getenv("RFX_RTC_DUMP")) {
            char fn[512];
            snprintf(fn, sizeof(fn), "%s/%s_%zx.hip", dump, name, std::hash<std::string>()(src));
            if (FILE *f = fopen(fn, "w")) {
                fwrite(src.data(), 1, src.size(), f);
                fclose(f);
            }
        }
    }
    std::string code;
    if (times == 1 && read_code(path, code)) { // compiled by an earlier process (or by the build step): milliseconds
        hipFunction_t fn = load(code, name);
        if (fn) {
            g_disk_loads++;
            (*g_cache)[sig] = fn;
            if (trace()) fprintf(stderr, "[rfx] rtc: %s loaded from %s\n", what, path.c_str());
            return fn;
        }
    }
    // Compiling takes seconds: only a plan that comes back, over enough rows for the faster kernel to matter, is worth it.  The
    // first occurrence (and any small input) runs the prebuilt kernel.  RFX_RTC_EAGER=1: compile at first sight (tests).
    if (!getenv("RFX_RTC_EAGER") && (times < 2 || nrows < (1LL << 24))) return NULL;
    if (trace()) fprintf(stderr, "[rfx] rtc: compiling %s ...\n", what);
    hipFunction_t fn = NULL;
    if (compile(src, code)) {
        fn = load(code, name);
        if (fn) write_code(path, code);
    }
    (*g_cache)[sig] = fn;
    if (trace()) fprintf(stderr, "[rfx] rtc: %s\n", fn ? "ready" : "not available for this plan: the prebuilt kernel runs");
    return fn;
}

// K1 for one plan: rfx_scalar_kernel.hpp's body with the plan's descriptors as a constexpr (every kind / column / operator test
// folds away).  *na_stride = accumulator slots per workgroup in ws (k_filter_aggr_final reads them).
static void filter_aggr_text(const Plan &P, std::string &sig, std::string &src) {
    bool deep = false;
    for (int i = 0; i < P.nx; i++) deep |= P.xs[i].nops > 1;
    const int nc = P.ncols < 1 ? 1 : P.ncols;
    int u = nc <= 4 ? 4 : (nc <= 6 ? 2 : 1);
    if (getenv("RFX_FA_U")) u = atoi(getenv("RFX_FA_U")) >= 4 ? 4 : (atoi(getenv("RFX_FA_U")) >= 2 ? 2 : 1); // development: row pairs per lane and tile
    char head[256];
    snprintf(head, sizeof(head), "#define FA_NC %d\n#define FA_NA %d\n#define FA_U %d\n#define FA_NP %d\n#define FA_NX %d\n#define FA_DEEP %s\n", nc, P.nagg, u, P.npred, P.nx,
             deep ? "true" : "false");
    std::string cond;
    plan_text(P, NULL, cond);
    sig = std::string(head) + cond;
    src = sig +
                            "#include \"rfx_scalar_kernel.hpp\"\n"
                            "extern \"C\" __global__ __launch_bounds__(RFX_BLOCK) void k_filter_aggr_plan(const Plan P0, Acc *__restrict__ ws) {\n"
                            "    constexpr Plan D = RTC_PLAN;\n"
                            "    filter_aggr_body<FA_NC, FA_NA, FA_U, FA_NP, FA_NX, FA_DEEP>(D, P0, ws);\n"
                            "}\n";
}

int rfx_rtc_filter_aggr(rfx_ctx *c, const Plan &P, int grid, void *ws, int *na_stride) {
    if (P.nagg < 1 || (c->flags & RFX_TUNE_NO_RTC) || !rtc_ready()) return RFX_ESTATE;
    std::string sig, src;
    filter_aggr_text(P, sig, src);
    hipFunction_t fn = plan_kernel(c->device, sig, src, "k_filter_aggr_plan", P.nrows, "a fused filter + aggregate kernel for this plan");
    if (!fn) return RFX_ESTATE;
    Plan Pv = P;
    void *wsv = ws;
    void *args[] = {&Pv, &wsv};
    RFX_HIP_CHECK(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, RFX_BLOCK, 1, 1, 0, c->stream, args, NULL));
    __atomic_fetch_add(&g_launches, 1, __ATOMIC_RELAXED);
    *na_stride = P.nagg + 1;
    return RFX_OK;
}

int rfx_rtc_group_few(rfx_ctx *c, const Plan &P, const GroupArgs &G, int grid) {
    if (G.range < 1 || G.range > RFX_FEW_MAX_GROUPS || P.nagg < 1 || P.nrows >= (1LL << 32)) return RFX_ESTATE;
    if (!rtc_ready()) return RFX_ESTATE;
    const int npt = P.npred == 0 ? 0 : (P.npred <= 2 ? 2 : RFX_MAX_PREDS);
    int u = P.ncols <= 4 ? 2 : 1;
    if (getenv("RFX_FEW_U")) u = atoi(getenv("RFX_FEW_U")) >= 2 ? 2 : 1; // development: rows per lane and tile
    char head[256];
    snprintf(head, sizeof(head), "#define FEW_NC %d\n#define FEW_NA %d\n#define FEW_NG %d\n#define FEW_NPT %d\n#define FEW_U %d\n#define FEW_FMA %d\n#define FEW_PREFETCH %d\n", P.ncols, P.nagg, (int)G.range, npt, u,
             getenv("RFX_FEW_NO_FMA") ? atoi(getenv("RFX_FEW_NO_FMA")) : 1, // development: 0 = selects instead of the masked fma
             getenv("RFX_FEW_PREFETCH") ? atoi(getenv("RFX_FEW_PREFETCH")) : 1); // development: 0 = no next-tile prefetch (rounds 2-5)
    std::string cond;
    plan_text(P, &G, cond);
    const std::string sig = std::string(head) + cond;
    const std::string src = std::string(head) + cond + "#include \"rfx_group_few_rtc.hpp\"\n";
    hipFunction_t fn = plan_kernel(c->device, sig, src, "k_group_few", P.nrows, "a register-accumulator group-by kernel for this plan");
    if (!fn) return RFX_ESTATE;
    Plan Pv = P;
    GroupArgs Gv = G;
    void *args[] = {&Pv, &Gv};
    RFX_HIP_CHECK(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, RFX_BLOCK, 1, 1, 0, c->stream, args, NULL));
    __atomic_fetch_add(&g_launches, 1, __ATOMIC_RELAXED);
    return RFX_OK;
}

// K3's one-pass `where` for one plan: rfx_where_once_kernel.hpp's body with the comparisons as constants (the prebuilt instantiations decode
// operator, domain and conversions per predicate at run time: 96 registers and scratch; one i64 comparison as a plan: 69, none).
static void where_once_text(const Plan &P, std::string &sig, std::string &src) {
    const int nc = P.ncols < 1 ? 1 : P.ncols;
    char head[128];
    snprintf(head, sizeof(head), "#define WOP_NC %d\n#define WOP_NP %d\n", nc, P.npred < 1 ? 1 : P.npred);
    std::string cond;
    plan_text(P, NULL, cond);
    sig = std::string("//where_once\n") + head + cond;
    src = sig +
          "#include \"rfx_where_once_kernel.hpp\"\n"
          "extern \"C\" __global__ __launch_bounds__(WO_T) __attribute__((amdgpu_waves_per_eu(WOP_NC <= 2 ? 5 : 3))) void k_where_once_plan(const Plan P0, const WoArgs A) {\n"
          "    constexpr Plan D = RTC_PLAN;\n"
          "    where_once_body<WOP_NC, WOP_NP>(D, P0, A);\n"
          "}\n";
}
int rfx_rtc_where_once(rfx_ctx *c, const Plan &P, const WoArgs &A, int grid) {
    if (P.npred < 1 || P.npred > 4 || P.ncols > 4 || (c->flags & RFX_TUNE_NO_RTC) || !rtc_ready()) return RFX_ESTATE;
    std::string sig, src;
    where_once_text(P, sig, src);
    hipFunction_t fn = plan_kernel(c->device, sig, src, "k_where_once_plan", P.nrows, "a one-pass where kernel for this plan");
    if (!fn) return RFX_ESTATE;
    Plan Pv = P;
    WoArgs Av = A;
    void *args[] = {&Pv, &Av};
    RFX_HIP_CHECK(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, WO_T, 1, 1, 0, c->stream, args, NULL));
    __atomic_fetch_add(&g_launches, 1, __ATOMIC_RELAXED);
    return RFX_OK;
}
extern "C" int rfx_hip_rtc_prewarm_where(const rfx_pred_t *preds, int npred, int logic) {
    Plan P;
    int rc = rfx_plan_build(&P, preds, npred, logic, NULL, 0, NULL, NULL, 1, 0);
    if (rc != RFX_OK) return rc;
    if (P.npred < 1 || P.npred > 4 || P.ncols > 4 || !rtc_ready() || R.cache_dir.empty()) return RFX_ESTATE;
    std::string sig, src, code;
    where_once_text(P, sig, src);
    const std::string path = cache_path(src);
    if (read_code(path, code)) return RFX_OK;
    if (!compile(src, code)) return RFX_ESTATE;
    write_code(path, code);
    return read_code(path, code) ? RFX_OK : RFX_ESTATE;
}

// Compile the K1 kernel of a plan INTO THE DISK CACHE without a device (hiprtc needs none for an explicit --offload-arch): the build
// step's pre-warm, also usable by a deployment for its own recurring queries.  Column pointers in preds / aggs only tell columns
// apart (any distinct non-NULL values).  RFX_OK: the code object is on disk (already, or now); RFX_ESTATE: no compiler / no cache
// directory / the plan does not compile.
extern "C" int rfx_hip_rtc_prewarm_filter_aggr(const rfx_pred_t *preds, int npred, int logic, const rfx_agg_t *aggs, int nagg) {
    Plan P;
    int rc = rfx_plan_build(&P, preds, npred, logic, aggs, nagg, NULL, NULL, 1, 0);
    if (rc != RFX_OK) return rc;
    if (P.nagg < 1 || !rtc_ready() || R.cache_dir.empty()) return RFX_ESTATE;
    std::string sig, src, code;
    filter_aggr_text(P, sig, src);
    const std::string path = cache_path(src);
    if (read_code(path, code)) return RFX_OK;
    if (!compile(src, code)) return RFX_ESTATE;
    write_code(path, code);
    return read_code(path, code) ? RFX_OK : RFX_ESTATE;
}
