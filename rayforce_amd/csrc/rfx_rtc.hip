// rfx_rtc.hip -- kernels compiled at run time for ONE plan (hiprtc), cached by plan signature.
//
// The prebuilt kernels decode their plan (which column feeds which aggregate, of which kind, under which predicates) from
// descriptors at run time and must keep per-group state where a run-time index can reach it (LDS).  A kernel generated for one
// plan has every descriptor as a constant and can keep per-(aggregate, group) state in registers: rfx_group_few_rtc.hpp.
// libhiprtc.so is loaded on first use (dlopen: nothing links against it, a box without it simply keeps the prebuilt kernels), the
// generated text #includes the library's own kernel headers from the source tree next to librfx.so (<dir of librfx.so>/csrc and
// <dir>/../include).  A plan whose kernel cannot be built (no compiler, no sources, a compile error) runs the prebuilt kernel:
// RFX_ESTATE from here means exactly that and nothing else.  First use of a plan costs the compilation (seconds); RFX_TRACE=1 says so.
#include <dlfcn.h>
#include <stdlib.h>
#include <sys/stat.h>
#include <map>
#include <string>
#include "rfx_group_common.hpp"

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
    std::string inc_csrc, inc_api;
} R;
static std::map<std::string, hipFunction_t> *g_cache; // signature -> kernel (NULL: failed once, never retried)
static long long g_launches, g_compiles;
extern "C" void rfx_hip_rtc_stats(int64_t *launches, int64_t *compiles) {
    if (launches) *launches = g_launches;
    if (compiles) *compiles = g_compiles;
}

static bool trace() { return getenv("RFX_TRACE") != NULL; }
static bool rtc_ready() {
    if (R.state) return R.state > 0;
    R.state = -1;
    if (getenv("RFX_NO_RTC")) return false;
    Dl_info di;
    if (!dladdr((const void *)&rtc_ready, &di) || !di.dli_fname) return false;
    std::string so(di.dli_fname);
    const size_t cut = so.find_last_of('/');
    const std::string dir = cut == std::string::npos ? std::string(".") : so.substr(0, cut);
    R.inc_csrc = dir + "/csrc";
    R.inc_api = dir + "/../include";
    struct stat st;
    if (stat((R.inc_csrc + "/rfx_group_few_rtc.hpp").c_str(), &st) != 0 || stat((R.inc_api + "/rfx_hip.h").c_str(), &st) != 0) {
        if (trace()) fprintf(stderr, "[rfx] rtc: kernel sources not found beside %s\n", so.c_str());
        return false;
    }
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
    if (!R.create || !R.compile || !R.log_size || !R.log || !R.code_size || !R.code || !R.destroy) return false;
    R.state = 1;
    return true;
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
        ADD("{ %d, %d, %d, %d, %d, %d, %s, %d }, ", P.preds[i].col, P.preds[i].rhs_col, P.preds[i].op, P.preds[i].dom_f64, P.preds[i].lhs_cvt, P.preds[i].rhs_cvt,
            cnan ? "0x7FF8000000000000ULL" : "0ULL", P.preds[i].more);
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

static hipFunction_t build(const std::string &src, const char *name) {
    rtcProgram prog = NULL;
    if (R.create(&prog, src.c_str(), "rfx_plan_kernel.hip", 0, NULL, NULL) != 0) return NULL;
    const std::string i1 = "-I" + R.inc_csrc, i2 = "-I" + R.inc_api;
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-fast-math", "-ffp-contract=off", i1.c_str(), i2.c_str()};
    const int rc = R.compile(prog, (int)(sizeof(opts) / sizeof(opts[0])), opts);
    if (rc != 0) {
        size_t ls = 0;
        R.log_size(prog, &ls);
        std::string log(ls + 1, 0);
        if (ls) R.log(prog, &log[0]);
        if (trace()) fprintf(stderr, "[rfx] rtc: compile failed (%d): %.2000s\n", rc, log.c_str());
        R.destroy(&prog);
        return NULL;
    }
    size_t cs = 0;
    R.code_size(prog, &cs);
    std::string code(cs, 0);
    R.code(prog, &code[0]);
    R.destroy(&prog);
    hipModule_t mod;
    hipFunction_t fn = NULL;
    if (hipModuleLoadData(&mod, code.data()) != hipSuccess || hipModuleGetFunction(&fn, mod, name) != hipSuccess) {
        if (trace()) fprintf(stderr, "[rfx] rtc: module load failed\n");
        return NULL;
    }
    return fn; // (the module lives as long as the process: a handful of plans)
}

// the kernel of a generated text: from the cache, or compiled now if the plan has come back often enough over enough rows
static hipFunction_t plan_kernel(const std::string &sig, const std::string &src, const char *name, i64 nrows, const char *what) {
    if (!g_cache) g_cache = new std::map<std::string, hipFunction_t>();
    auto it = g_cache->find(sig);
    if (it != g_cache->end()) return it->second;
    // Compiling takes seconds: only a plan that comes back, over enough rows for the faster kernel to matter, is worth it.  The
    // first occurrence (and any small input) runs the prebuilt kernel.  RFX_RTC_EAGER=1: compile at first sight (tests).
    static std::map<std::string, int> *seen;
    if (!seen) seen = new std::map<std::string, int>();
    const int times = ++(*seen)[sig];
    if (!getenv("RFX_RTC_EAGER") && (times < 2 || nrows < (1LL << 24))) return NULL;
    if (trace()) fprintf(stderr, "[rfx] rtc: compiling %s ...\n", what);
    hipFunction_t fn = build(src, name);
    g_compiles++;
    (*g_cache)[sig] = fn;
    if (trace()) fprintf(stderr, "[rfx] rtc: %s\n", fn ? "ready" : "not available for this plan: the prebuilt kernel runs");
    return fn;
}

// K1 for one plan: rfx_scalar_kernel.hpp's body with the plan's descriptors as a constexpr (every kind / column / operator test
// folds away).  *na_stride = accumulator slots per workgroup in ws (k_filter_aggr_final reads them).
int rfx_rtc_filter_aggr(rfx_ctx *c, const Plan &P, int grid, void *ws, int *na_stride) {
    if (P.nagg < 1 || (c->flags & RFX_TUNE_NO_RTC) || !rtc_ready()) return RFX_ESTATE;
    bool deep = false;
    for (int i = 0; i < P.nx; i++) deep |= P.xs[i].nops > 1;
    const int nc = P.ncols < 1 ? 1 : P.ncols;
    const int u = nc <= 4 ? 4 : (nc <= 6 ? 2 : 1);
    char head[256];
    snprintf(head, sizeof(head), "#define FA_NC %d\n#define FA_NA %d\n#define FA_U %d\n#define FA_NP %d\n#define FA_NX %d\n#define FA_DEEP %s\n", nc, P.nagg, u, P.npred, P.nx,
             deep ? "true" : "false");
    std::string cond;
    plan_text(P, NULL, cond);
    const std::string sig = std::string(head) + cond;
    const std::string src = sig +
                            "#include \"rfx_scalar_kernel.hpp\"\n"
                            "extern \"C\" __global__ __launch_bounds__(RFX_BLOCK) void k_filter_aggr_plan(const Plan P0, Acc *__restrict__ ws) {\n"
                            "    constexpr Plan D = RTC_PLAN;\n"
                            "    filter_aggr_body<FA_NC, FA_NA, FA_U, FA_NP, FA_NX, FA_DEEP>(D, P0, ws);\n"
                            "}\n";
    hipFunction_t fn = plan_kernel(sig, src, "k_filter_aggr_plan", P.nrows, "a fused filter + aggregate kernel for this plan");
    if (!fn) return RFX_ESTATE;
    Plan Pv = P;
    void *wsv = ws;
    void *args[] = {&Pv, &wsv};
    RFX_HIP_CHECK(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, RFX_BLOCK, 1, 1, 0, c->stream, args, NULL));
    g_launches++;
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
    snprintf(head, sizeof(head), "#define FEW_NC %d\n#define FEW_NA %d\n#define FEW_NG %d\n#define FEW_NPT %d\n#define FEW_U %d\n#define FEW_FMA %d\n", P.ncols, P.nagg, (int)G.range, npt, u,
             getenv("RFX_FEW_NO_FMA") ? atoi(getenv("RFX_FEW_NO_FMA")) : 1); // development: 0 = selects instead of the masked fma
    std::string cond;
    plan_text(P, &G, cond);
    const std::string sig = std::string(head) + cond;
    const std::string src = std::string(head) + cond + "#include \"rfx_group_few_rtc.hpp\"\n";
    hipFunction_t fn = plan_kernel(sig, src, "k_group_few", P.nrows, "a register-accumulator group-by kernel for this plan");
    if (!fn) return RFX_ESTATE;
    Plan Pv = P;
    GroupArgs Gv = G;
    void *args[] = {&Pv, &Gv};
    RFX_HIP_CHECK(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, RFX_BLOCK, 1, 1, 0, c->stream, args, NULL));
    g_launches++;
    return RFX_OK;
}
