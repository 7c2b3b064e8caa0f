// jh_bcast.hip -- BlockArray broadcast (src/Jets.jl:889-911) for ANY elementwise expression, fused into one pass.
//
// The reference's `copyto!(dest::BlockArray, bc::Broadcasted{BlockArrayStyle})` walks the blocks and runs Julia's
// compiled broadcast kernel per block, so `d .= exp.(a .* u) .+ v ./ w` is one loop over k+1 streams whatever the
// expression is.  The device equivalent of "Julia compiles the expression" is hiprtc: the host language prints the
// Broadcasted tree as a C expression over x0..x{k-1} (vector elements) and s0..s{m-1} (scalars); it is compiled ONCE for
// gfx950 into a kernel that streams the slabs with 16-byte loads, one pack per lane (the shape that streams fastest on
// this chip, jh_vecops.hip), with -ffp-contract=off so every operation is rounded as written (a*u + b*v gives the bits of
// the reference's broadcast, not an FMA's).  The BlockArray is one slab, so block boundaries do not exist for a broadcast.
#include "jh_internal.h"

#include <hip/hiprtc.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

struct jh_bcast {
    int dtype = JH_F32;
    int nvec = 0, nscal = 0;
    int wide_mask = 0;                  // bit k: scalar k is Float64-based in a 32-bit program (JH_SCALAR_WIDE): it enters the expression as a double
    int real_mask = 0;                  // bit k: vector operand k is REAL in a complex program (a real mask on a complex vector); bit nvec + k: scalar k is
    // a program is device-agnostic (one code object for gfx950); its module is loaded per DEVICE on first use there
    struct on_device {
        hipModule_t module = nullptr;
        hipFunction_t fn_vec = nullptr;     // 16 bytes per lane (every operand 16-byte aligned)
        hipFunction_t fn_scalar = nullptr;  // one element per lane (views at odd offsets)
        hipFunction_t fn_batched = nullptr; // fn_vec over MANY equally long destinations at once (blockIdx.y = item; pointers from a table)
    };
    static constexpr int MAX_DEV = 32;
    mutable on_device dev[MAX_DEV];
    std::vector<char> code;
    std::string expr;
};

namespace {

constexpr int JH_BCAST_MAX_VEC = 8, JH_BCAST_MAX_SCAL = 8;

// complex arithmetic for the generated code: plain formulas, every product and sum rounded (like jh_vecops.hip)
const char *k_prelude = R"SRC(
// the precision of a mixed operation is the wider operand's (Julia's promotion == C++'s usual arithmetic conversions for float / double / ints)
template <typename A, typename B> struct wd_ { typedef decltype(A() + B()) t; };
template <typename B> struct is_num_ { static const bool v = false; };
template <> struct is_num_<float> { static const bool v = true; };
template <> struct is_num_<double> { static const bool v = true; };
template <> struct is_num_<int> { static const bool v = true; };
template <> struct is_num_<long> { static const bool v = true; };
template <> struct is_num_<long long> { static const bool v = true; };
template <bool C, typename T> struct en_ {};
template <typename T> struct en_<true, T> { typedef T t; };
template <typename R> struct cx {
    typedef R real_t;
    R re, im;
    __device__ cx() : re(0), im(0) {}
    __device__ cx(R r) : re(r), im(0) {}
    __device__ cx(R r, R i) : re(r), im(i) {}
    template <typename Q> __device__ explicit cx(cx<Q> o) : re((R)o.re), im((R)o.im) {}   // the store's conversion (and a widening)
};
#define CXW_ typename wd_<R1, R2>::t
template <typename R1, typename R2> __device__ inline cx<CXW_> operator+(cx<R1> a, cx<R2> b) { typedef CXW_ W; return cx<W>((W)a.re + (W)b.re, (W)a.im + (W)b.im); }
template <typename R1, typename R2> __device__ inline cx<CXW_> operator-(cx<R1> a, cx<R2> b) { typedef CXW_ W; return cx<W>((W)a.re - (W)b.re, (W)a.im - (W)b.im); }
template <typename R> __device__ inline cx<R> operator-(cx<R> a) { return cx<R>(-a.re, -a.im); }
template <typename R1, typename R2> __device__ inline cx<CXW_> operator*(cx<R1> a, cx<R2> b)
{
    typedef CXW_ W;
    return cx<W>((W)a.re * (W)b.re - (W)a.im * (W)b.im, (W)a.re * (W)b.im + (W)a.im * (W)b.re);
}
template <typename R1, typename R2> __device__ inline cx<CXW_> operator/(cx<R1> a, cx<R2> b)
{
    typedef CXW_ W;
    const W den = (W)b.re * (W)b.re + (W)b.im * (W)b.im;
    return cx<W>(((W)a.re * (W)b.re + (W)a.im * (W)b.im) / den, ((W)a.im * (W)b.re - (W)a.re * (W)b.im) / den);
}
#undef CXW_
// real (x) complex, part by part (Julia's a::Real * z): any arithmetic type; an integer takes the complex operand's precision, a double widens it
#define CXR_ typename en_<is_num_<B>::v, cx<typename wd_<R, B>::t>>::t
template <typename R, typename B> __device__ inline CXR_ operator+(cx<R> a, B b) { typedef typename wd_<R, B>::t W; return cx<W>((W)a.re + (W)b, (W)a.im); }
template <typename R, typename B> __device__ inline CXR_ operator+(B a, cx<R> b) { typedef typename wd_<R, B>::t W; return cx<W>((W)a + (W)b.re, (W)b.im); }
template <typename R, typename B> __device__ inline CXR_ operator-(cx<R> a, B b) { typedef typename wd_<R, B>::t W; return cx<W>((W)a.re - (W)b, (W)a.im); }
template <typename R, typename B> __device__ inline CXR_ operator-(B a, cx<R> b) { typedef typename wd_<R, B>::t W; return cx<W>((W)a - (W)b.re, -(W)b.im); }
template <typename R, typename B> __device__ inline CXR_ operator*(cx<R> a, B b) { typedef typename wd_<R, B>::t W; return cx<W>((W)a.re * (W)b, (W)a.im * (W)b); }
template <typename R, typename B> __device__ inline CXR_ operator*(B a, cx<R> b) { typedef typename wd_<R, B>::t W; return cx<W>((W)a * (W)b.re, (W)a * (W)b.im); }
template <typename R, typename B> __device__ inline CXR_ operator/(cx<R> a, B b) { typedef typename wd_<R, B>::t W; return cx<W>((W)a.re / (W)b, (W)a.im / (W)b); }
template <typename R, typename B> __device__ inline CXR_ operator/(B a, cx<R> b) { typedef typename wd_<R, B>::t W; return cx<W>((W)a) / b; }
#undef CXR_
template <typename R> __device__ inline cx<R> conj(cx<R> a) { return cx<R>(a.re, -a.im); }
template <typename R> __device__ inline R real(cx<R> a) { return a.re; }
template <typename R> __device__ inline R imag(cx<R> a) { return a.im; }
template <typename R> __device__ inline R abs2(cx<R> a) { return a.re * a.re + a.im * a.im; }
template <typename R> __device__ inline R abs(cx<R> a) { return hypot(a.re, a.im); }
template <typename R> __device__ inline cx<R> exp(cx<R> a) { const R e = exp(a.re); return cx<R>(e * cos(a.im), e * sin(a.im)); }
__device__ inline float conj(float a) { return a; }
__device__ inline double conj(double a) { return a; }
__device__ inline float real(float a) { return a; }
__device__ inline double real(double a) { return a; }
__device__ inline float imag(float) { return 0.f; }
__device__ inline double imag(double) { return 0.0; }
__device__ inline float abs2(float a) { return a * a; }
__device__ inline double abs2(double a) { return a * a; }
__device__ inline float sign(float a) { return (a > 0.f) - (a < 0.f); }
__device__ inline double sign(double a) { return (a > 0.0) - (a < 0.0); }
)SRC";

std::string build_source(const std::string &expr, int dtype, int nvec, int nscal, int real_mask, int wide_mask)
{
    const bool is64 = (dtype == JH_F64 || dtype == JH_C64), cplx = jh_dtype_complex(dtype);
    const char *R = is64 ? "double" : "float";
    const int NS = is64 ? 2 : 4;                       // scalars per 16-byte pack
    const int E = cplx ? 2 : 1;                        // scalars per element
    std::string s = k_prelude;
    s += std::string("typedef ") + R + " R;\n";
    s += cplx ? "typedef cx<R> T;\n" : "typedef R T;\n";
    // (round 5, session 3) packs are ADDRESSED through a type aligned like the scalar -- the same global_load / global_store_dwordx4, which gfx950 executes at
    // any dword-aligned address -- and a vector whose length is not a whole number of packs loads its last pack from n - NS and stores only the scalars
    // that pack's lane owns: a BlockArray of 255^3-element blocks (odd length, views off the 16-byte grid) keeps the 16-byte-per-lane kernel
    const std::string A = is64 ? "8" : "4";
    s += "typedef R V __attribute__((ext_vector_type(" + std::to_string(NS) + ")));\n";
    s += "typedef V UV_ __attribute__((aligned(" + A + ")));\n";
    s += "typedef const UV_ __attribute__((address_space(1))) *gvp;\ntypedef UV_ __attribute__((address_space(1))) *gvq;\n";
    s += "typedef const R __attribute__((address_space(1))) *gsp;\ntypedef R __attribute__((address_space(1))) *gsq;\n";
    // a REAL operand of a complex program (src/Jets.jl:899-904 pairs blocks whatever their eltypes; Julia's real (x) complex
    // arithmetic = the prelude's mixed operators): NS/2 reals per lane where the complex operands have NS/2 elements
    if (cplx && NS / 2 > 1) s += "typedef R RV __attribute__((ext_vector_type(" + std::to_string(NS / 2) + ")));\ntypedef RV URV_ __attribute__((aligned(" + A + ")));\n#define RGET(P, e) ((R)(P)[e])\n";
    else s += "typedef R RV;\ntypedef R URV_;\n#define RGET(P, e) ((R)(P))\n";
    s += "typedef const URV_ __attribute__((address_space(1))) *grvp;\n";
    auto is_real = [&](int k) { return cplx && ((real_mask >> k) & 1); };
    // element accessors on a pack
    if (!cplx) {
        s += "#define GET(P, e) ((T)(P)[e])\n#define PUT(P, e, val) (P)[e] = (R)(val)\n";
    } else {
        s += "#define GET(P, e) T((P)[2 * (e)], (P)[2 * (e) + 1])\n#define PUT(P, e, val) do { T t_ = (val); (P)[2 * (e)] = t_.re; (P)[2 * (e) + 1] = t_.im; } while (0)\n";
    }
    std::string params = "R *dst_";   // may alias an operand (x .= f.(x, y)): no __restrict__
    for (int k = 0; k < nvec; k++) params += ", const R *p" + std::to_string(k);
    // a WIDE scalar (JH_SCALAR_WIDE: Julia's Float64 against 32-bit elements) enters as a double: every operation it meets is then a double
    // operation by the language's own promotion, and the element type comes back with the ONE rounding of the final conversion to T
    auto is_wide = [&](int k) { return !is64 && ((wide_mask >> k) & 1); };
    auto sreal = [&](int k) { return std::string(is_wide(k) ? "double" : "R"); };
    for (int k = 0; k < nscal; k++) params += ", " + sreal(k) + " sr" + std::to_string(k) + ", " + sreal(k) + " si" + std::to_string(k);
    params += ", long n_scalars";
    std::string scal;
    for (int k = 0; k < nscal; k++) {
        const std::string i = std::to_string(k);
        // a REAL scalar of a complex program (bit nvec + k of real_mask): Julia's `a::Real * z` works part by part (the prelude's
        // mixed operators) -- no 0 * Inf from an imaginary part the scalar does not have
        if (cplx && !is_real(nvec + k)) scal += "    const cx<" + sreal(k) + "> s" + i + "(sr" + i + ", si" + i + ");\n";
        else scal += "    const " + sreal(k) + " s" + i + " = sr" + i + "; (void)si" + i + ";\n";
    }
    // ---- 16 bytes per lane
    s += "extern \"C\" __global__ __launch_bounds__(256) void jh_bcast_vec(" + params + ")\n{\n" + scal;
    const std::string NSs = std::to_string(NS);
    // pack v nominally starts at scalar s_ = v * NS and is loaded from c_: s_, or n - NS for the last, partial pack (n_scalars >= NS: the host checks)
    const std::string pack_pos = "        const long s_ = v * " + NSs + ", c_ = s_ + " + NSs + " <= n_scalars ? s_ : n_scalars - " + NSs + ";\n";
    const std::string pack_store = "        if (c_ == s_) __builtin_nontemporal_store(r_, (gvq)(dst_ + c_));\n"
                                   "        else {\n#pragma unroll\n            for (int e = 0; e < " + NSs + "; e++) if (c_ + e >= s_) ((gsq)dst_)[c_ + e] = r_[e];\n        }\n";
    s += "    const long nvec = (n_scalars + " + NSs + " - 1) / " + NSs + ";\n";
    s += "    const long stride = (long)gridDim.x * 256;\n";
    s += "    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += stride) {\n" + pack_pos;
    for (int k = 0; k < nvec; k++)
        s += is_real(k) ? "        const RV X" + std::to_string(k) + " = __builtin_nontemporal_load((grvp)(p" + std::to_string(k) + " + c_ / 2));\n"
                        : "        const V X" + std::to_string(k) + " = __builtin_nontemporal_load((gvp)(p" + std::to_string(k) + " + c_));\n";
    s += "        V r_;\n#pragma unroll\n        for (int e = 0; e < " + std::to_string(NS / E) + "; e++) {\n";
    for (int k = 0; k < nvec; k++)
        s += is_real(k) ? "            const R x" + std::to_string(k) + " = RGET(X" + std::to_string(k) + ", e);\n"
                        : "            const T x" + std::to_string(k) + " = GET(X" + std::to_string(k) + ", e);\n";
    s += "            const T val_ = (T)(" + expr + ");\n            PUT(r_, e, val_);\n        }\n";
    s += pack_store + "    }\n}\n";
    // ---- 16 bytes per lane, batched: item blockIdx.y takes its destination, operands and scalars from device tables
    // (jh_bcast_apply_many: the children of a tall nonlinear operator in ONE launch instead of one launch per child)
    // item_fast_: the ITEM is the fastest block index (workgroups that read the same pack of a shared operand -- the model
    // vector every child of a tall nonlinear operator evaluates -- are dispatched together: it comes from L2, not HBM)
    const bool any_wide = !is64 && wide_mask != 0;          // then the scalar table holds doubles for every scalar of the program
    s += std::string("extern \"C\" __global__ __launch_bounds__(256) void jh_bcast_vec_batched(const void *const *tbl_, const ") + (any_wide ? "double" : "R") +
         " *sc_, long n_scalars, int item_fast_, int shared_mask_)\n{\n";
    // item_fast_ > 1: COLUMN bands of that many tiles (late round 4; the tall forward's walk): blockIdx.x = (item, tile within the band),
    // blockIdx.y = band -- the workgroups of one item stream item_fast_ consecutive tiles, the shared operand's band is re-used by every item from L2
    s += "    const long band_ = item_fast_ > 1 ? item_fast_ : 1;\n";
    s += "    const long item_ = item_fast_ ? blockIdx.x / band_ : blockIdx.y;\n";
    s += "    const long tile_ = item_fast_ ? (long)blockIdx.y * band_ + blockIdx.x % band_ : blockIdx.x;\n";
    s += "    const long ntile_ = item_fast_ ? (long)gridDim.y * band_ : gridDim.x;\n";
    s += "    const void *const *row_ = tbl_ + item_ * " + std::to_string(nvec + 1) + ";\n";
    s += "    R *dst_ = (R *)row_[0];\n";
    for (int k = 0; k < nvec; k++) s += "    const R *p" + std::to_string(k) + " = (const R *)row_[" + std::to_string(k + 1) + "];\n";
    if (nscal > 0) s += std::string("    const ") + (any_wide ? "double" : "R") + " *srow_ = sc_ + item_ * " + std::to_string(2 * nscal) + ";\n";
    for (int k = 0; k < nscal; k++) {
        const std::string i = std::to_string(k);
        s += "    const " + sreal(k) + " sr" + i + " = (" + sreal(k) + ")srow_[" + std::to_string(2 * k) + "], si" + i + " = (" + sreal(k) + ")srow_[" + std::to_string(2 * k + 1) + "];\n";
    }
    s += scal;
    s += "    const long nvec = (n_scalars + " + NSs + " - 1) / " + NSs + ";\n";
    s += "    const long stride = ntile_ * 256;\n";
    s += "    for (long v = tile_ * 256 + threadIdx.x; v < nvec; v += stride) {\n" + pack_pos;
    // an operand every item shares is loaded through the caches (bit k of shared_mask_), the streamed ones nontemporally
    for (int k = 0; k < nvec; k++)
        s += std::string("        const ") + (is_real(k) ? "RV" : "V") + " X" + std::to_string(k) + " = ((shared_mask_ >> " + std::to_string(k) + ") & 1) ? *((" +
             (is_real(k) ? "grvp" : "gvp") + ")(p" + std::to_string(k) + (is_real(k) ? " + c_ / 2" : " + c_") + ")) : __builtin_nontemporal_load((" + (is_real(k) ? "grvp" : "gvp") + ")(p" +
             std::to_string(k) + (is_real(k) ? " + c_ / 2" : " + c_") + "));\n";
    s += "        V r_;\n#pragma unroll\n        for (int e = 0; e < " + std::to_string(NS / E) + "; e++) {\n";
    for (int k = 0; k < nvec; k++)
        s += is_real(k) ? "            const R x" + std::to_string(k) + " = RGET(X" + std::to_string(k) + ", e);\n"
                        : "            const T x" + std::to_string(k) + " = GET(X" + std::to_string(k) + ", e);\n";
    s += "            const T val_ = (T)(" + expr + ");\n            PUT(r_, e, val_);\n        }\n";
    s += pack_store + "    }\n}\n";
    // ---- one element per lane
    s += "extern \"C\" __global__ __launch_bounds__(256) void jh_bcast_scalar(" + params + ")\n{\n" + scal;
    s += "    const long nel = n_scalars / " + std::to_string(E) + ";\n";
    s += "    const long stride = (long)gridDim.x * 256;\n";
    s += "    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nel; i += stride) {\n";
    for (int k = 0; k < nvec; k++) {
        const std::string i = std::to_string(k);
        if (is_real(k)) s += "        const R x" + i + " = ((gsp)p" + i + ")[i];\n";
        else
        s += cplx ? "        const T x" + i + "(((gsp)p" + i + ")[2 * i], ((gsp)p" + i + ")[2 * i + 1]);\n"
                  : "        const T x" + i + " = ((gsp)p" + i + ")[i];\n";
    }
    s += "        const T val_ = (T)(" + expr + ");\n";
    s += cplx ? "        ((gsq)dst_)[2 * i] = val_.re; ((gsq)dst_)[2 * i + 1] = val_.im;\n" : "        ((gsq)dst_)[i] = val_;\n";
    s += "    }\n}\n";
    return s;
}

std::mutex g_cache_mutex;
std::map<std::string, jh_bcast *> g_cache;      // (dtype, nvec, nscal, expr) -> compiled program, shared by every handle

int compile_code(const std::string &expr, int dtype, int nvec, int nscal, std::vector<char> &code, int real_mask = 0, int wide_mask = 0)
{
    const std::string src = build_source(expr, dtype, nvec, nscal, real_mask, wide_mask);
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, src.c_str(), "jh_bcast.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
        return jh_fail(JH_ERR_HIP, "jh_bcast_compile: hiprtcCreateProgram failed");
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-std=c++17"};
    const hiprtcResult rc = hiprtcCompileProgram(prog, 5, opts);
    if (rc != HIPRTC_SUCCESS) {
        size_t n = 0;
        std::string log;
        if (hiprtcGetProgramLogSize(prog, &n) == HIPRTC_SUCCESS && n > 1) {
            log.resize(n);
            (void)hiprtcGetProgramLog(prog, &log[0]);
        }
        (void)hiprtcDestroyProgram(&prog);
        if (log.size() > 1500) log.resize(1500);
        return jh_fail(JH_ERR_INVALID, "jh_bcast_compile: `%s` does not compile as an elementwise expression over x0..x%d, s0..s%d:\n%s",
                       expr.c_str(), nvec - 1, nscal - 1, log.c_str());
    }
    size_t code_size = 0;
    if (hiprtcGetCodeSize(prog, &code_size) != HIPRTC_SUCCESS || code_size == 0) {
        (void)hiprtcDestroyProgram(&prog);
        return jh_fail(JH_ERR_HIP, "jh_bcast_compile: no code object");
    }
    code.resize(code_size);
    (void)hiprtcGetCode(prog, code.data());
    (void)hiprtcDestroyProgram(&prog);
    return JH_OK;
}

std::mutex g_load_mutex;
// the program's functions on the CURRENT context's device
int loaded(const jh_bcast *bc, const jh_bcast::on_device **out)
{
    const int d = jh_ctx().device;
    JH_REQUIRE(d >= 0 && d < jh_bcast::MAX_DEV, "jh_bcast: device %d beyond the %d this build keeps modules for", d, jh_bcast::MAX_DEV);
    jh_bcast::on_device &f = bc->dev[d];
    // double-checked: fn_batched is published LAST (release) and read with acquire, so a host thread that sees it also sees the
    // module and the other two functions another thread loaded
    if (!__atomic_load_n(&f.fn_batched, __ATOMIC_ACQUIRE)) {
        std::lock_guard<std::mutex> lock(g_load_mutex);
        if (!f.fn_batched) {
            hipModule_t mod = nullptr;
            hipFunction_t fv = nullptr, fs = nullptr, fb = nullptr;
            hipError_t e = hipModuleLoadData(&mod, bc->code.data());
            if (e == hipSuccess) e = hipModuleGetFunction(&fv, mod, "jh_bcast_vec");
            if (e == hipSuccess) e = hipModuleGetFunction(&fs, mod, "jh_bcast_scalar");
            if (e == hipSuccess) e = hipModuleGetFunction(&fb, mod, "jh_bcast_vec_batched");
            if (e != hipSuccess) {
                if (mod) (void)hipModuleUnload(mod);
                return jh_fail(JH_ERR_HIP, "jh_bcast: loading the code object on device %d: %s", d, hipGetErrorString(e));
            }
            f.module = mod;
            f.fn_vec = fv;
            f.fn_scalar = fs;
            __atomic_store_n(&f.fn_batched, fb, __ATOMIC_RELEASE);
        }
    }
    *out = &f;
    return JH_OK;
}

int compile(const std::string &expr, int dtype, int nvec, int nscal, jh_bcast **out, int real_mask = 0, int wide_mask = 0)
{
    std::vector<char> code;
    JH_TRY(compile_code(expr, dtype, nvec, nscal, code, real_mask, wide_mask));
    jh_bcast *bc = new jh_bcast();
    bc->dtype = dtype;
    bc->nvec = nvec;
    bc->nscal = nscal;
    bc->real_mask = real_mask;
    bc->wide_mask = wide_mask;
    bc->expr = expr;
    bc->code.swap(code);
    const jh_bcast::on_device *fns = nullptr;
    const int st = loaded(bc, &fns);                      // load it on the current device right away: errors surface at compile time
    if (st != JH_OK) { delete bc; return st; }
    *out = bc;
    return JH_OK;
}

}  // namespace

extern "C" {

static int check_request(const char *expr, int dtype, int nvec, int nscal)
{
    JH_REQUIRE(expr, "jh_bcast_compile: null expression");
    JH_REQUIRE(jh_dtype_size(dtype) != 0, "jh_bcast_compile: unknown dtype %d", dtype);
    JH_REQUIRE(nvec >= 0 && nvec <= JH_BCAST_MAX_VEC && nscal >= 0 && nscal <= JH_BCAST_MAX_SCAL,
               "jh_bcast_compile: at most %d vector and %d scalar operands (got %d, %d)", JH_BCAST_MAX_VEC, JH_BCAST_MAX_SCAL, nvec, nscal);
    JH_REQUIRE(strlen(expr) > 0 && strlen(expr) < 4096, "jh_bcast_compile: empty or oversized expression");
    return JH_OK;
}

int jh_bcast_check(const char *expr, int dtype, int nvec, int nscal)
{
    JH_TRY(check_request(expr, dtype, nvec, nscal));
    std::vector<char> code;
    return compile_code(expr, dtype, nvec, nscal, code);      // hiprtc cross-compiles: no device needed
}

int jh_bcast_check_typed(const char *expr, int dtype, int nvec, int real_mask, int nscal, int wide_mask)
{
    JH_TRY(check_request(expr, dtype, nvec, nscal));
    std::vector<char> code;
    return compile_code(expr, dtype, nvec, nscal, code, real_mask, (dtype == JH_F64 || dtype == JH_C64) ? 0 : wide_mask);
}

int jh_bcast_compile(const char *expr, int dtype, int nvec, int nscal, jh_bcast **out)
{
    JH_TRY(jh_require_ready());
    JH_REQUIRE(out, "jh_bcast_compile: null argument");
    JH_TRY(check_request(expr, dtype, nvec, nscal));
    const std::string key = std::to_string(dtype) + "/" + std::to_string(nvec) + "/" + std::to_string(nscal) + "/" + expr;
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    auto it = g_cache.find(key);
    if (it != g_cache.end()) {
        *out = it->second;
        return JH_OK;
    }
    jh_bcast *bc = nullptr;
    JH_TRY(compile(expr, dtype, nvec, nscal, &bc));
    g_cache[key] = bc;
    *out = bc;
    return JH_OK;
}

int jh_bcast_compile_typed(const char *expr, int dtype, int nvec, int real_mask, int nscal, int wide_mask, jh_bcast **out)
{
    JH_TRY(jh_require_ready());
    JH_REQUIRE(out, "jh_bcast_compile_typed: null argument");
    JH_TRY(check_request(expr, dtype, nvec, nscal));
    JH_REQUIRE(nvec + nscal < 31, "jh_bcast_compile_typed: at most 30 operands and scalars");
    JH_REQUIRE(real_mask >= 0 && real_mask < (1 << (nvec + nscal > 0 ? nvec + nscal : 1)), "jh_bcast_compile_typed: real_mask has bits beyond the %d operands and %d scalars", nvec, nscal);
    JH_REQUIRE(real_mask == 0 || jh_dtype_complex(dtype), "jh_bcast_compile_typed: real operands only make a difference in a complex program");
    JH_REQUIRE(wide_mask >= 0 && wide_mask < (1 << (nscal > 0 ? nscal : 1)), "jh_bcast_compile_typed: wide_mask has bits beyond the %d scalars", nscal);
    if (dtype == JH_F64 || dtype == JH_C64) wide_mask = 0;              // nothing is wider than 64-bit elements
    if (real_mask == 0 && wide_mask == 0) return jh_bcast_compile(expr, dtype, nvec, nscal, out);
    const std::string key = std::to_string(dtype) + "/" + std::to_string(nvec) + "/" + std::to_string(nscal) + "/r" + std::to_string(real_mask) + "/w" +
                            std::to_string(wide_mask) + "/" + expr;
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    auto it = g_cache.find(key);
    if (it != g_cache.end()) {
        *out = it->second;
        return JH_OK;
    }
    jh_bcast *bc = nullptr;
    JH_TRY(compile(expr, dtype, nvec, nscal, &bc, real_mask, wide_mask));
    g_cache[key] = bc;
    *out = bc;
    return JH_OK;
}

int jh_bcast_compile_mixed(const char *expr, int dtype, int nvec, int real_mask, int nscal, jh_bcast **out)
{
    return jh_bcast_compile_typed(expr, dtype, nvec, real_mask, nscal, 0, out);
}

int jh_bcast_apply(const jh_bcast *bc, jh_bvec *dst, const jh_bvec *const *x, const double *scal_re_im)
{
    JH_TRY(jh_enter(dst));
    JH_REQUIRE(bc && dst, "jh_bcast_apply: null argument");
    JH_REQUIRE(bc->nvec == 0 || x, "jh_bcast_apply: null operand list");
    JH_REQUIRE(bc->nscal == 0 || scal_re_im, "jh_bcast_apply: null scalar list");
    JH_REQUIRE(dst->dtype == bc->dtype, "jh_bcast_apply: destination dtype %d, program compiled for %d", dst->dtype, bc->dtype);
    uintptr_t bits = (uintptr_t)dst->data;
    for (int k = 0; k < bc->nvec; k++) {
        JH_REQUIRE(x[k], "jh_bcast_apply: operand %d is null", k);
        JH_REQUIRE(x[k]->ctx == dst->ctx, "jh_bcast_apply: operand %d lives in context %d, the destination in %d", k, x[k]->ctx, dst->ctx);
        const int want_dt = ((bc->real_mask >> k) & 1) ? (bc->dtype == JH_C32 ? JH_F32 : JH_F64) : bc->dtype;      // a real operand of a complex program
        JH_REQUIRE(x[k]->dtype == want_dt, "jh_bcast_apply: operand %d has dtype %d, program compiled for %d", k, x[k]->dtype, want_dt);
        JH_REQUIRE(x[k]->length == dst->length, "jh_bcast_apply: operand %d has %lld elements, destination %lld (DimensionMismatch)", k,
                   (long long)x[k]->length, (long long)dst->length);
        bits |= (uintptr_t)x[k]->data;
    }
    if (dst->length == 0) return JH_OK;
    const bool cplx = jh_dtype_complex(bc->dtype), is64 = (bc->dtype == JH_F64 || bc->dtype == JH_C64);
    const int64_t n_scalars = dst->length * (cplx ? 2 : 1);
    const int NS = is64 ? 2 : 4;
    const bool vec_ok = ((bits & (uintptr_t)(is64 ? 7u : 3u)) == 0) && n_scalars >= NS;   // aligned like the scalar, at least one pack (the kernel's packs are under-aligned, its last one partial)
    // kernel arguments: dst, p0.., (sr, si).., n_scalars
    void *dptr = dst->data;
    const void *ptrs[JH_BCAST_MAX_VEC];
    float sf[2 * JH_BCAST_MAX_SCAL];
    double sd[2 * JH_BCAST_MAX_SCAL];
    long n_arg = (long)n_scalars;
    void *args[2 + JH_BCAST_MAX_VEC + 2 * JH_BCAST_MAX_SCAL];
    int na = 0;
    args[na++] = &dptr;
    for (int k = 0; k < bc->nvec; k++) { ptrs[k] = x[k]->data; args[na++] = &ptrs[k]; }
    for (int k = 0; k < 2 * bc->nscal; k++) {
        if (is64 || ((bc->wide_mask >> (k / 2)) & 1)) { sd[k] = scal_re_im[k]; args[na++] = &sd[k]; }   // (a wide scalar of a 32-bit program is a double parameter)
        else { sf[k] = (float)scal_re_im[k]; args[na++] = &sf[k]; }
    }
    args[na++] = &n_arg;
    const int64_t work = vec_ok ? (n_scalars + NS - 1) / NS : dst->length;
    int64_t grid = (work + 255) / 256;
    if (grid > ((int64_t)1 << 23)) grid = (int64_t)1 << 23;              // grid x 256 threads < 2^32; the kernel strides
    const jh_bcast::on_device *fns = nullptr;
    JH_TRY(loaded(bc, &fns));
    JH_CHECK_HIP(hipModuleLaunchKernel(vec_ok ? fns->fn_vec : fns->fn_scalar, (unsigned)grid, 1, 1, 256, 1, 1, 0, jh_ctx().stream, args, nullptr));
    return JH_OK;
}

// the launches of a batch whose device tables are in place (first use or a repeat of the same batch)
static int launch_batched_groups(jh_context &c, const std::vector<jh_context::bcast_group> &groups, const void *dev, size_t tbl_bytes)
{
    hipStream_t st = c.stream;
    for (const jh_context::bcast_group &g : groups) {
        const jh_bcast *bc = g.bc;
        const bool cplx = jh_dtype_complex(bc->dtype);
        const int64_t n_scalars = g.len * (cplx ? 2 : 1);
        const size_t row = (size_t)bc->nvec + 1, sc_row = 2 * (size_t)bc->nscal * (g.wide_scal ? 8 : 4);
        const jh_bcast::on_device *fns = nullptr;
        JH_TRY(loaded(bc, &fns));
        int item_fast = g.item_fast, shared_mask = g.shared_mask;
        const int64_t xmul = item_fast > 1 ? item_fast : 1;
        for (int k0 = 0; k0 < g.gcount; k0 += 65535) {
            const int gy = g.gcount - k0 < 65535 ? g.gcount - k0 : 65535;
            const void *tbl_arg = (const char *)dev + (g.tbl_at + (size_t)k0 * row) * sizeof(void *);
            const void *sc_arg = (const char *)dev + tbl_bytes + g.sc_at + (size_t)k0 * sc_row;
            long n_arg = (long)n_scalars;
            void *args[5] = {&tbl_arg, &sc_arg, &n_arg, &item_fast, &shared_mask};
            if (item_fast)
                JH_CHECK_HIP(hipModuleLaunchKernel(fns->fn_batched, (unsigned)(gy * xmul), (unsigned)g.gx, 1, 256, 1, 1, 0, st, args, nullptr));
            else
                JH_CHECK_HIP(hipModuleLaunchKernel(fns->fn_batched, (unsigned)g.gx, (unsigned)gy, 1, 256, 1, 1, 0, st, args, nullptr));
        }
    }
    return JH_OK;
}

// Items that share a program and a length, with every pointer on 16 bytes, run as ONE launch over (packs, items) with device
// tables -- provided no operand overlaps ANOTHER item's destination (then the items have no order among them and the batch
// may be regrouped by program).  Returns JH_OK with *done = false when the request is not of that shape (the caller then
// launches item by item, in order).
static int apply_many_batched(int count, const jh_bcast *const *progs, jh_bvec *const *dsts, const jh_bvec *const *xs, const double *scal_re_im,
                              bool *done)
{
    *done = false;
    if (count < 4) return JH_OK;
    // ---- the SAME batch as last time?  A tall nonlinear operator evaluates F(m) into the same vectors again and again: its device tables are
    // kept (in a buffer of the context's own) while the argument arrays are element for element those of the last batched call, no vector
    // handle has been destroyed since (a handle's data never changes otherwise) and the knobs are the same -- then nothing is
    // validated, grouped or copied again (16 384 children of 16 KiB: 1.3 ms per call for a 0.1 ms kernel before)
    if (progs[0] && dsts[0]) {
        jh_context *cc = jh_ctx_by_id(dsts[0]->ctx);
        if (cc && cc->bcast_last.count == count && cc->bcast_last.gen == jh_bvec_generation.load() &&
            cc->bcast_last.knob_item == cc->bcast_item_fast && cc->bcast_last.knob_band == cc->bcast_band) {
            const jh_context::bcast_batch &L = cc->bcast_last;
            const size_t nx = L.key.size() - 2 * (size_t)count;
            if (std::memcmp(L.key.data(), progs, sizeof(void *) * (size_t)count) == 0 &&
                std::memcmp(L.key.data() + count, dsts, sizeof(void *) * (size_t)count) == 0 &&
                (nx == 0 || (xs && std::memcmp(L.key.data() + 2 * (size_t)count, xs, sizeof(void *) * nx) == 0)) &&
                (L.scal.empty() || (scal_re_im && std::memcmp(L.scal.data(), scal_re_im, sizeof(double) * L.scal.size()) == 0))) {
                JH_TRY(jh_enter(dsts[0]));
                JH_TRY(launch_batched_groups(jh_ctx(), L.groups, L.dev, L.tbl_bytes));
                *done = true;
                return JH_OK;
            }
        }
    }
    // ---- every item well-formed and 16-byte aligned?  (the item-by-item path reports errors)
    std::vector<int64_t> xoff((size_t)count), soff((size_t)count);
    int64_t ix = 0, is = 0;
    uintptr_t bits = 0;
    for (int k = 0; k < count; k++) {
        const jh_bcast *bc = progs[k];
        if (!bc || !dsts[k] || dsts[k]->dtype != bc->dtype || dsts[k]->length == 0) return JH_OK;
        if (dsts[k]->ctx != dsts[0]->ctx) return JH_OK;                      // (the item-by-item path works context by context)
        if ((bc->nvec > 0 && !xs) || (bc->nscal > 0 && !scal_re_im)) return JH_OK;
        const int NS = (bc->dtype == JH_F64 || bc->dtype == JH_C64) ? 2 : 4;
        if ((dsts[k]->length * (jh_dtype_complex(bc->dtype) ? 2 : 1)) < NS) return JH_OK;      // (less than one pack: item by item, the scalar kernel)
        xoff[(size_t)k] = ix;
        soff[(size_t)k] = is;
        bits = (uintptr_t)dsts[k]->data;
        for (int j = 0; j < bc->nvec; j++) {
            const jh_bvec *x = xs[ix + j];
            if (!x || x->dtype != bc->dtype || x->length != dsts[k]->length || x->ctx != dsts[k]->ctx) return JH_OK;
            bits |= (uintptr_t)x->data;
        }
        // aligned like the scalar (device arrays of the element type are; a raw wrapped pointer need not be): the kernel's packs are under-aligned
        if (bits & (uintptr_t)(NS == 2 ? 7u : 3u)) return JH_OK;
        ix += bc->nvec;
        is += 2 * bc->nscal;
    }
    {   // one launch has no order between items: an operand must not overlap ANOTHER item's destination (its own is fine, elementwise)
        std::vector<std::pair<uintptr_t, int>> dst_lo((size_t)count);
        auto nbytes = [&](int k) { return (size_t)dsts[k]->length * jh_dtype_size(dsts[k]->dtype); };
        for (int k = 0; k < count; k++) dst_lo[(size_t)k] = {(uintptr_t)dsts[k]->data, k};
        std::sort(dst_lo.begin(), dst_lo.end());
        for (size_t k = 1; k < dst_lo.size(); k++)
            if (dst_lo[k].first < dst_lo[k - 1].first + nbytes(dst_lo[k - 1].second)) return JH_OK;      // overlapping destinations
        for (int k = 0; k < count; k++)
            for (int j = 0; j < progs[k]->nvec; j++) {
                const uintptr_t lo = (uintptr_t)xs[xoff[(size_t)k] + j]->data, hi = lo + nbytes(k);
                auto it = std::lower_bound(dst_lo.begin(), dst_lo.end(), std::make_pair(lo, -1));
                if (it != dst_lo.end() && it->first < hi && !(it->second == k && it->first == lo)) return JH_OK;
                if (it != dst_lo.begin()) {
                    --it;
                    if (it->first + nbytes(it->second) > lo && !(it->second == k && it->first == lo)) return JH_OK;
                }
            }
    }
    JH_TRY(jh_enter(dsts[0]));
    // ---- group by (program, length), keeping first-appearance order
    struct Group { const jh_bcast *bc; int64_t len; std::vector<int> items; size_t tbl_at = 0, sc_at = 0; };
    std::vector<Group> groups;
    std::map<std::pair<const jh_bcast *, int64_t>, size_t> where;
    for (int k = 0; k < count; k++) {
        const auto key = std::make_pair(progs[k], dsts[k]->length);
        auto it = where.find(key);
        if (it == where.end()) {
            where[key] = groups.size();
            Group g;
            g.bc = progs[k];
            g.len = dsts[k]->length;
            groups.push_back(g);
            it = where.find(key);
        }
        groups[it->second].items.push_back(k);
    }
    if (groups.size() * 2 > (size_t)count) return JH_OK;       // hardly anything to batch
    // ---- one pointer table and one scalar table for all groups (scalars as raw bytes: float or double per group)
    std::vector<const void *> tbl;
    std::vector<char> sc;
    for (Group &g : groups) {
        const bool is64 = (g.bc->dtype == JH_F64 || g.bc->dtype == JH_C64) || g.bc->wide_mask != 0;   // (a program with a wide scalar reads doubles)
        g.tbl_at = tbl.size();
        sc.resize((sc.size() + 15) / 16 * 16);
        g.sc_at = sc.size();
        for (int k : g.items) {
            tbl.push_back(dsts[k]->data);
            for (int j = 0; j < g.bc->nvec; j++) tbl.push_back(xs[xoff[(size_t)k] + j]->data);
            for (int q = 0; q < 2 * g.bc->nscal; q++) {
                const double v = scal_re_im[soff[(size_t)k] + q];
                if (is64) { const char *p = (const char *)&v; sc.insert(sc.end(), p, p + 8); }
                else { const float f = (float)v; const char *p = (const char *)&f; sc.insert(sc.end(), p, p + 4); }
            }
        }
    }
    const size_t tbl_bytes = (tbl.size() * sizeof(void *) + 255) / 256 * 256;
    // the tables live in a buffer of the context's own (not the shared scratch: they stay valid for the next call of the same batch)
    jh_context &c = jh_ctx();
    jh_context::bcast_batch &L = c.bcast_last;
    L.count = -1;                                             // (nothing to re-use until this batch is complete)
    hipStream_t st = c.stream;
    if (L.dev_cap < tbl_bytes + sc.size() + 16) {
        if (L.dev) { JH_CHECK_HIP(hipStreamSynchronize(st)); JH_CHECK_HIP(hipFree(L.dev)); L.dev = nullptr; L.dev_cap = 0; }
        size_t cap = (size_t)1 << 16;
        while (cap < tbl_bytes + sc.size() + 16) cap *= 2;
        JH_CHECK_HIP(jh_device_malloc(c.device, &L.dev, cap));
        L.dev_cap = cap;
    }
    void *dev = L.dev;
    JH_CHECK_HIP(hipMemcpyAsync(dev, tbl.data(), tbl.size() * sizeof(void *), hipMemcpyHostToDevice, st));
    if (!sc.empty()) JH_CHECK_HIP(hipMemcpyAsync((char *)dev + tbl_bytes, sc.data(), sc.size(), hipMemcpyHostToDevice, st));
    JH_CHECK_HIP(hipStreamSynchronize(st));                   // the staging vectors die at return (the copies are tiny)
    const int64_t knob = c.bcast_item_fast;                    // -1 automatic, 0 never, 1 whenever an operand is shared
    L.groups.clear();
    for (const Group &g : groups) {
        const jh_bcast *bc = g.bc;
        const bool cplx = jh_dtype_complex(bc->dtype), is64 = (bc->dtype == JH_F64 || bc->dtype == JH_C64);
        const int NS = is64 ? 2 : 4;
        const int64_t n_scalars = g.len * (cplx ? 2 : 1);
        const int gcount = (int)g.items.size();
        int64_t gx = ((n_scalars + NS - 1) / NS + 255) / 256;
        if (gx > 65535) gx = 65535;                           // the kernel strides
        // an operand every item of the group shares (the model vector of F(m) / point!) is loaded through the caches; beyond
        // 32 MiB per vector the items also become the fastest block index, so that it is read from HBM once per XCD instead of
        // once per item (256 children of 64 MiB: F(m) 4.8 -> 2.75 ms)
        int item_fast = 0, shared_mask = 0;
        for (int j = 0; j < bc->nvec; j++) {
            bool shared = gcount > 1;
            for (int q = 1; q < gcount && shared; q++)
                if (xs[xoff[(size_t)g.items[(size_t)q]] + j]->data != xs[xoff[(size_t)g.items[0]] + j]->data) shared = false;
            if (shared) shared_mask |= 1 << j;
        }
        if (shared_mask && knob != 0 && (knob == 1 || (size_t)g.len * jh_dtype_size(bc->dtype) >= ((size_t)32 << 20))) item_fast = 1;
        // ... in column bands of 32 tiles (128 KiB of every item at a time; knob bcast_band: tiles per band, 1 = the item-fastest order of round 3)
        const int64_t band = c.bcast_band > 0 ? c.bcast_band : 32;
        if (item_fast && band > 1 && gx >= 2 * band) {
            item_fast = (int)band;
            gx = (gx + band - 1) / band;                       // bands (blockIdx.y); blockIdx.x = item * band + tile within the band
        }
        if (knob == 0) shared_mask = 0;                        // A/B: the plain kernel
        L.groups.push_back(jh_context::bcast_group{bc, g.len, gx, gcount, item_fast, shared_mask, g.tbl_at, g.sc_at, is64 || bc->wide_mask != 0});
    }
    L.tbl_bytes = tbl_bytes;
    JH_TRY(launch_batched_groups(c, L.groups, dev, tbl_bytes));
    // remember the argument arrays: the next call with the same ones (and no vector handle born or gone meanwhile) goes straight to the launches
    L.key.assign((const void *const *)progs, (const void *const *)progs + count);
    L.key.insert(L.key.end(), (const void *const *)dsts, (const void *const *)dsts + count);
    if (ix > 0) L.key.insert(L.key.end(), (const void *const *)xs, (const void *const *)xs + ix);
    L.scal.assign(scal_re_im ? scal_re_im : nullptr, scal_re_im ? scal_re_im + is : nullptr);
    L.gen = jh_bvec_generation.load();
    L.knob_item = c.bcast_item_fast;
    L.knob_band = c.bcast_band;
    L.count = count;
    *done = true;
    return JH_OK;
}

int jh_bcast_apply_many(int count, const jh_bcast *const *progs, jh_bvec *const *dsts, const jh_bvec *const *xs, const double *scal_re_im)
{
    JH_REQUIRE(count >= 0 && (count == 0 || (progs && dsts)), "jh_bcast_apply_many: null argument");
    if (count > 0) {
        bool done = false;
        JH_TRY(apply_many_batched(count, progs, dsts, xs, scal_re_im, &done));
        if (done) return JH_OK;
    }
    int64_t ix = 0, is = 0;                                   // running offsets into the flattened operand / scalar lists
    for (int k = 0; k < count; k++) {
        JH_REQUIRE(progs[k], "jh_bcast_apply_many: program %d is null", k);
        JH_TRY(jh_bcast_apply(progs[k], dsts[k], xs ? xs + ix : nullptr, scal_re_im ? scal_re_im + is : nullptr));
        ix += progs[k]->nvec;
        is += 2 * progs[k]->nscal;
    }
    return JH_OK;
}

int jh_bcast_destroy(jh_bcast *bc)
{
    (void)bc;   // programs live in the process-wide cache (one per distinct expression) until jh_shutdown
    return JH_OK;
}

}  // extern "C"

void jh_bcast_clear_cache()
{
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    for (auto &kv : g_cache) {
        for (auto &f : kv.second->dev)
            if (f.module) (void)hipModuleUnload(f.module);
        delete kv.second;
    }
    g_cache.clear();
}
