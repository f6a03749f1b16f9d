// C-ABI host runtime of libfrp.so: handle, weight/program blob, gallery snapshots, the
// detect -> align -> embed -> match pipeline on one HIP stream, per-stage HIP-event timing.
// Interface and the reference call sites each entry point replaces: include/frp.h.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types only: the library opens librccl at first use (frp_dist_*)
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include <thread>
#include <atomic>

#include "frp.h"
#ifdef FRP_LAB
#include "frp_lab.h"
#endif
#include "frp_blob.h"
#include "frp_internal.h"
#include "jpeg_host.h"

using namespace frp;

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct TensorDims {
    int h = 0, w = 0, c = 0;
    bool f32 = false;
    bool f8 = false;      // OCP E4M3 bytes (BASELINE config 5: fp8 matrix path of the embedder)
};

struct Net {
    std::vector<frp_conv_op> ops;
    int n_bufs = 0;
    int in_buf = 0, in_ch = 0;
    std::vector<DevBuf> bufs;
    std::vector<TensorDims> dims;   // per physical buffer, for the last planned shape
    std::vector<int64_t> wino_off;  // per op: byte offset of its Winograd weight image in the data section, or -1
    // K-concat (conv_mfma.hip): a block's 1x1 stride-s shortcut conv folded into its 3x3 stride-s conv as a second K segment
    std::vector<int> kc_skip;       // per op: 1 = a shortcut conv that its consumer computes (not launched)
    std::vector<int> kc_src;        // per op: index of the shortcut op folded into this conv, or -1
    std::vector<int64_t> kc_w_off, kc_bias_off;   // per consumer op: concatenated weights [Cout][9 Cin + Cin2] / summed bias [Cout]
};

enum { EV_START = 0, EV_H2D, EV_PRE, EV_DET, EV_DEC, EV_ALIGN, EV_EMB, EV_L2, EV_MATCH, EV_D2H, EV_COUNT };

// A network pass as a captured hipGraph (round 5): the ~50 / ~85 launches of a detector / embedder pass replayed by ONE call when the
// same pass - same program, shapes, buffers, operands, switches - is asked for again (run_net).  What the host side of the pass does
// besides launching (counters, the FC's split-K bookkeeping, the planned dims) is recorded with it and re-applied on replay.
struct NetGraph {
    std::string key;
    hipGraphExec_t exec = nullptr;
    uint64_t epoch = 0;                 // frp_handle::alloc_epoch at capture: any (re)allocation or weight load since makes it stale
    double dflops = 0, df8flops = 0;
    int64_t dlaunches = 0, df8launches = 0;
    int fc_ksplit = 0, fc_ktot = 0;
    const float* fc_bias = nullptr;
    std::vector<TensorDims> dims;
};

}  // namespace

struct frp_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    std::string err;
    frp_config cfg{};
    // weights
    bool have_weights = false;
    frp_blob_header hdr{};
    DevBuf wdata;
    Net det, emb;
    // resident frames (tightly packed u8 [B,H,W,3])
    DevBuf frames;
    int rB = 0, rH = 0, rW = 0;
    int n_cu = 256;                   // compute units of the device (queried once at create)
    // overlapped ingest: the NEXT batch is copied on its own stream while the current one is processed
    DevBuf frames_next;
    int nB = 0, nH = 0, nW = 0;
    bool next_valid = false;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_next_ready = nullptr, ev_next_free = nullptr;
    std::vector<void*> pinned;       // frp_host_alloc blocks, freed with the handle
    // detector source: the resident frames, or a resized copy of them (pyramid scales)
    DevBuf scaled;
    int dH = 0, dW = 0;              // dims of the detector source
    bool det_scaled = false;
    int canvas_h = 0, canvas_w = 0;
    int det_op_limit = -1;           // >= 0: frp_debug_det_prefix - the detector program stops behind this many ops
    // captured passes (run_net): graphs, the keys seen once (a pass is captured the SECOND time it is asked for: the first allocates and
    // sets kernel attributes), the keys whose capture failed, the allocation epoch
    std::vector<NetGraph> graphs;
    std::vector<std::string> graph_seen, graph_bad;
    uint64_t alloc_epoch = 1;
    int64_t graph_replays = 0;
    // multi-GPU (frp_dist_init): this handle's RCCL communicator, rank and world size
    void* comm = nullptr;
    int dist_rank = 0, dist_world = 0;
    DevBuf det_hashes;               // frp_debug_det_hashes: one 64-bit hash per detector op, taken right behind the op
    bool det_hash_on = false;
    // per-call results (device)
    DevBuf boxes, kps, scores, counts, anchor, face_slot, nfaces, q16, part_cos, part_idx, best_cos, best_idx, scratch, splitk_ws, dense_logits;
    int fc_ksplit = 0;               // >0: the embedder's FC wrote split-K slabs; l2norm reduces them (-1: factor chosen on the device)
    int fc_ktot = 0;
    const float* fc_bias = nullptr;
    int last_B = 0, last_K = 0, last_nfaces = 0;   // last_nfaces -1: count still on the device (resolve_count)
    int last_cap = 0, pend_cap = 0;
    double pend_flops = 0.0, pend_f8flops = 0.0;
    bool ev_pending = false;   // frp_process_resident's stage events are recorded but not yet read (see settle_events)
    bool last_matched = false;
    int32_t* h_nfaces = nullptr;   // pinned
    unsigned char* pin_stage = nullptr;   // pinned staging of the result fetch
    size_t pin_cap = 0;
    // gallery snapshot
    DevBuf gallery;
    int64_t g_rows = 0;
    DevBuf g_reserved;               // frp_gallery_reserve: filled by the caller, swapped in by frp_gallery_commit
    // JPEG ingest (frp_upload_jpeg_async): page-locked coefficient staging, device coefficients / tables / sample planes
    // (two staging buffers in turn: the host decodes batch t+1 while the copy of batch t still reads the other one)
    void* jpeg_pin[2] = {nullptr, nullptr};
    size_t jpeg_pin_cap[2] = {0, 0};
    int jpeg_turn = 0;
    DevBuf jpeg_coef, jpeg_planes;
    int64_t ctr_jpeg_device_batches = 0;   // batches whose entropy decode ran on the device (frp_debug_jpeg_device_batches)
    DevBuf jpeg_scan, jpeg_err;      // device entropy decode (restart-interval streams): compressed scans + interval offsets + tables; per-image error flags
    hipEvent_t ev_jpeg_h2d[2] = {nullptr, nullptr};     // the copy out of jpeg_pin[i] has finished
    bool jpeg_h2d_pending[2] = {false, false};
    // exact compat rows (frp_gallery_exact): float64 [g_rows x 512] as enrolled, next to the unit fp16 snapshot
    bool g_exact = false;
    DevBuf gx, gx_q, gx_out;
    // profiling
    hipEvent_t ev[EV_COUNT]{};
    frp_counters ctr{};
};

namespace {

int fail(frp_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    return code;
}

// librccl, opened at first use (frp_dist_*: the gallery all-gather; a process that never goes multi-GPU does not load it)
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) { r.err = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?"); return; }
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString) r.err = "librccl lacks an expected symbol";
    });
    return r;
}


#define HIPCHK(h, expr)                                                                              \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            return fail(h, FRP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));          \
    } while (0)

#define FRPCHK(expr)                 \
    do {                             \
        int _r = (expr);             \
        if (_r != FRP_OK) return _r; \
    } while (0)

int ensure(frp_handle* h, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap && b.p) return FRP_OK;
    if (b.p) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    const size_t want = std::max<size_t>(bytes, 256);
    ++h->alloc_epoch;                   // (captured passes hold device pointers)
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(h, FRP_ERR_OOM, std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e));
    }
    b.cap = want;
    return FRP_OK;
}

void release(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// OCP FP8 E4M3FN (bias 7, no infinities; S.1111.111 = NaN, decoded as 0 here: the packer never emits it)
float fp8_e4m3_value(unsigned char c) {
    const int e = (c >> 3) & 0xF, m = c & 7;
    float v;
    if (e == 15 && m == 7) v = 0.f;
    else if (e == 0) v = std::ldexp((float)m / 8.0f, -6);
    else v = std::ldexp(1.0f + (float)m / 8.0f, e - 7);
    return (c & 0x80) ? -v : v;
}

// fp32 -> fp16 bit pattern, round to nearest even (the rounding of numpy's astype(float16))
uint16_t f32_to_f16_bits(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));     // NaN / inf
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                    // rounds to inf (>= 65520)
    if (x < 0x33000001u) return (uint16_t)sign;                                                 // rounds to zero (<= 2^-25)
    const int exp = (int)(x >> 23) - 127;
    uint32_t mant = (x & 0x7fffffu) | 0x800000u;
    int shift;
    uint32_t base;
    if (exp < -14) { shift = 13 + (-14 - exp); base = 0; }                                      // subnormal half
    else { shift = 13; base = (uint32_t)(exp + 15) << 10; mant &= 0x7fffffu; }
    uint32_t q = mant >> shift;
    const uint32_t rem = mant & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) ++q;                                            // a carry walks into the exponent
    return (uint16_t)(sign | (base + q));
}

float f16_bits_to_f32(uint16_t hbits) {
    const uint32_t sign = (uint32_t)(hbits & 0x8000u) << 16;
    const int e = (hbits >> 10) & 31;
    const uint32_t m = hbits & 0x3ffu;
    float v;
    if (e == 0) v = std::ldexp((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = std::ldexp((float)(m | 0x400u), e - 25);
    uint32_t bits;
    memcpy(&bits, &v, 4);
    bits |= sign;
    memcpy(&v, &bits, 4);
    return v;
}

// Weight image of the Winograd kernel (conv3x3_wino.hip) from folded fp16 weights [Cout][3][3][Cin]: per (cout tile of
// 128, 64-channel block, kernel row, 16-channel slice) one 16 KiB stage = the LDS image itself: [frequency f][32-cout block]
// [32 x 16-byte slots] (couts beyond Cout zero), the 8-channel half h of cout r of a block at slot (2 r + h) ^ ((r >> 3) & 1):
// every fragment of a stage is one per-lane base + an immediate (conv3x3_wino.hip: wino_u_slot).  U = G g: g0, (g0+g1+g2)/2,
// (g0-g1+g2)/2, g2 - exact in fp32 on fp16 inputs, rounded once.
void build_wino_image(const uint16_t* w16, int Cin, int Cout, uint16_t* img) {
    const int cpt = Cin / 64, nct = (Cout + 127) / 128;
    for (int ct = 0; ct < nct; ++ct)
        for (int cb = 0; cb < cpt; ++cb)
            for (int kh = 0; kh < 3; ++kh)
                for (int kk = 0; kk < 4; ++kk) {
                    uint16_t* st = img + ((((size_t)ct * cpt + cb) * 3 + kh) * 4 + kk) * 8192;
                    for (int row = 0; row < 128; ++row) {
                        const int co = ct * 128 + row;
                        for (int hh = 0; hh < 2; ++hh)
                            for (int e = 0; e < 8; ++e) {
                                const int ci = cb * 64 + kk * 16 + hh * 8 + e;
                                float g[3] = {0.f, 0.f, 0.f};
                                if (co < Cout)
                                    for (int kw = 0; kw < 3; ++kw) g[kw] = f16_bits_to_f32(w16[(((size_t)co * 3 + kh) * 3 + kw) * Cin + ci]);
                                const float u[4] = {g[0], (g[0] + g[1] + g[2]) * 0.5f, (g[0] - g[1] + g[2]) * 0.5f, g[2]};
                                // stage layout (conv3x3_wino.hip: wino_u_slot): [f][32-cout block][slot (2 r + h) ^ ((r >> 3) & 1)][8 channels]
                                const int r = row & 31, slot = (2 * r + hh) ^ ((r >> 3) & 1);
                                for (int f = 0; f < 4; ++f) st[f * 2048 + (row >> 5) * 512 + slot * 8 + e] = f32_to_f16_bits(u[f]);
                            }
                    }
                }
}

float logit_threshold(float t) {
    if (!(t > 0.f)) return -INFINITY;
    if (t >= 1.f) return INFINITY;
    return (float)std::log((double)t / (1.0 - (double)t));
}

void rec(frp_handle* h, int which) {
    if (h->cfg.profile) (void)hipEventRecord(h->ev[which], h->stream);
}

// Plan + run one conv program.  in dims: [batch, H, W, in_ch] already written to bufs[in_buf].
int plan_net(frp_handle* h, Net& net, int batch, int H, int W, bool skip_input = false) {
    net.dims.assign(net.n_bufs, TensorDims());
    std::vector<size_t> need(net.n_bufs, 0);
    net.dims[net.in_buf] = {H, W, net.in_ch, false};
    need[net.in_buf] = skip_input ? 0 : (size_t)batch * H * W * net.in_ch * 2;
    for (const frp_conv_op& op : net.ops) {
        TensorDims in = net.dims[op.in_buf];
        if (in.c == 0) return fail(h, FRP_ERR_BLOB, "program reads an unwritten buffer");
        if (op.flags & FRP_FLAG_FLATTEN) in = {1, 1, in.h * in.w * in.c, false};
        if (in.c != op.cin) return fail(h, FRP_ERR_BLOB, "program channel mismatch");
        const int pad = op.ksize / 2;
        TensorDims out;
        out.h = (in.h + 2 * pad - op.ksize) / op.stride + 1;
        out.w = (in.w + 2 * pad - op.ksize) / op.stride + 1;
        out.c = op.cout;
        out.f32 = (op.flags & FRP_FLAG_OUT_F32) != 0;
        if (out.h <= 0 || out.w <= 0) return fail(h, FRP_ERR_INVALID, "input too small for the network");
        if (in.f32) return fail(h, FRP_ERR_BLOB, "program reads an fp32 tensor as a conv input");
        const bool op_f8 = (op.flags & FRP_OPFLAG_FP8_MFMA) != 0;
        if (in.f8 != op_f8) return fail(h, FRP_ERR_BLOB, "program operand precision mismatch (fp8 op <-> fp8 tensor)");
        if (op_f8 && !(op.ksize == 3 && op.stride == 1 && (op.cin & 127) == 0 && !(op.flags & (FRP_FLAG_OUT_F32 | FRP_FLAG_FLATTEN | FRP_FLAG_RES_UP2))))
            return fail(h, FRP_ERR_BLOB, "op shape not covered by the fp8 matrix path");
        out.f8 = (op.flags & FRP_OPFLAG_OUT_FP8) != 0;
        if (out.f8 && (out.f32 || !op_f8)) return fail(h, FRP_ERR_BLOB, "fp8 primary output needs an fp8 op");
        if (op.res_buf >= 0) {                     // the epilogue reads the residual unchecked: validate it here
            const TensorDims& r = net.dims[op.res_buf];
            const bool up2 = (op.flags & FRP_FLAG_RES_UP2) != 0;
            if (r.c != op.cout || r.f32 || r.f8 || (up2 ? (r.h * 2 != out.h || r.w * 2 != out.w) : (r.h != out.h || r.w != out.w)))
                return fail(h, FRP_ERR_BLOB, "program residual shape mismatch");
        }
        net.dims[op.out_buf] = out;
        need[op.out_buf] = std::max(need[op.out_buf], (size_t)batch * out.h * out.w * out.c * (out.f32 ? 4 : out.f8 ? 1 : 2));
        if (op.out2_buf >= 0) {                    // fp8 copy of an fp16 primary output
            if (out.f32 || out.f8) return fail(h, FRP_ERR_BLOB, "fp8 copy of a non-fp16 output");
            TensorDims o2 = out;
            o2.f8 = true;
            net.dims[op.out2_buf] = o2;
            need[op.out2_buf] = std::max(need[op.out2_buf], (size_t)batch * out.h * out.w * out.c);
        }
    }
    for (int i = 0; i < net.n_bufs; ++i)
        if (need[i]) FRPCHK(ensure(h, net.bufs[i], need[i]));
    return FRP_OK;
}

// first detector op as the fused u8 stem (no NHWC8 blob)?
bool stem_fusable(const Net& net) {
    if (net.ops.empty()) return false;
    const frp_conv_op& op = net.ops[0];
    return op.in_buf == net.in_buf && op.cin == 8 && op.cout == 32 && op.ksize == 3 && op.stride == 2 &&
           op.act == FRP_ACT_RELU && op.res_buf < 0 && (op.flags & ~0) == 0 && (op.real_ch & 0xffff) == 3;
}

// ... and the second one (3x3 s2 32->64 + ReLU) reading nothing but the first: both stems in one kernel
bool stem12_fusable(const Net& net) {
    if (!stem_fusable(net) || net.ops.size() < 2) return false;
    const frp_conv_op& a = net.ops[0];
    const frp_conv_op& b = net.ops[1];
    if (!(b.in_buf == a.out_buf && b.cin == 32 && b.cout == 64 && b.ksize == 3 && b.stride == 2 && b.act == FRP_ACT_RELU &&
          b.res_buf < 0 && b.flags == 0))
        return false;
    for (size_t i = 2; i < net.ops.size(); ++i)          // the stem1 map must have no other reader
        if (net.ops[i].in_buf == a.out_buf || net.ops[i].res_buf == a.out_buf) {
            // (physical buffers are recycled: a later tensor may live in the same buffer - only a read
            // before the next write of that buffer would be the stem1 map)
            bool rewritten = false;
            for (size_t j = 2; j < i; ++j) rewritten |= net.ops[j].out_buf == a.out_buf;
            if (!rewritten) return false;
        }
    return true;
}

// `n_dev`: the number of images that really exist lives in device memory (`batch` is then the capacity the buffers were
// planned for): every kernel derives its tile count from it.  The flop counters are charged for `batch` images and
// corrected by the caller once the count is known.
// `allow_wino` false: the direct kernels also where a Winograd weight image exists (calls of few faces, run_embed).
int run_net_body(frp_handle* h, Net& net, int batch, int H, int W, double* flops, int64_t* launches, const StemParams* stem,
                 const int32_t* n_dev, bool allow_wino) {
    // dims are re-derived while walking (physical buffers are reused by several tensors)
    std::vector<TensorDims> d(net.n_bufs);
    d[net.in_buf] = {H, W, net.in_ch, false};
    const char* wbase = (const char*)h->wdata.p;
    bool first = true;
    size_t skip = 0;
    if (&net == &h->det && h->det_hash_on) HIPCHK(h, hipMemsetAsync(h->det_hashes.p, 0, 64 * 8, h->stream));
    // tile class of the conv launches (conv_common.h: conv_small_m): FRP_SMALL_M=0 never quarter tiles, =1 always, unset: by
    // the tile count of each launch (A/B runs, tests that pin a kernel family)
    const char* sm_env = getenv("FRP_SMALL_M");
    const int small_m = !sm_env ? 0 : (sm_env[0] == '0' ? -1 : 1);
    const bool s2_optin = getenv("FRP_S2") != nullptr;        // the opt-in stride-2 row-patch kernel (conv3x3_s2.hip; A/B runs, its pipeline test)
    // both detector stems in one kernel (the stem1 map never reaches HBM); FRP_NO_FUSED_STEM12=1 keeps
    // stem1 (fused with the u8 normalisation) and stem2 (generic conv) apart for A/B runs
    if (stem && stem12_fusable(net) && (stem->Hc % 4) == 0 && (stem->Wc % 4) == 0 && !getenv("FRP_NO_FUSED_STEM12")) {
        const frp_conv_op& a = net.ops[0];
        const frp_conv_op& b = net.ops[1];
        Stem12Params sp{};
        sp.frames = stem->frames; sp.B = stem->B; sp.H = stem->H; sp.W = stem->W;
        sp.row_stride = stem->row_stride; sp.frame_stride = stem->frame_stride;
        sp.Hc = stem->Hc; sp.Wc = stem->Wc; sp.Ho1 = stem->Hc / 2; sp.Wo1 = stem->Wc / 2; sp.Ho2 = stem->Hc / 4; sp.Wo2 = stem->Wc / 4;
        sp.rgb_in = stem->rgb_in;
        sp.w1 = (const _Float16*)(wbase + a.w_off); sp.bias1 = (const float*)(wbase + a.bias_off);
        sp.w2 = (const _Float16*)(wbase + b.w_off); sp.bias2 = (const float*)(wbase + b.bias_off);
        sp.out = (_Float16*)net.bufs[b.out_buf].p;
        hipError_t e = launch_stem12_u8(sp, h->stream);
        if (e != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("launch_stem12_u8: ") + hipGetErrorString(e));
        d[a.out_buf] = {sp.Ho1, sp.Wo1, 32, false};
        d[b.out_buf] = {sp.Ho2, sp.Wo2, 64, false};
        *flops += 2.0 * batch * sp.Ho1 * sp.Wo1 * 9.0 * 3 * 32 + 2.0 * batch * sp.Ho2 * sp.Wo2 * 9.0 * 32 * 64;
        *launches += 1;
        if (h->det_hash_on) {
            e = launch_tensor_hash(sp.out, (size_t)batch * sp.Ho2 * sp.Wo2 * 64 * 2, (unsigned long long*)h->det_hashes.p + 1, h->stream);
            if (e != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("tensor_hash: ") + hipGetErrorString(e));
        }
        skip = 2;
        first = false;
    }
    // embedder stem (chips NHWC8 -> 3x3 s1 3->64 + PReLU): dedicated kernel; FRP_NO_EMB_STEM=1 keeps the generic one
    bool fuse_stem = false, fuse_even_only = false;
    size_t fuse_op = 0;
    EmbStemParams fused{};
    if (!stem && !net.ops.empty()) {
        const frp_conv_op& a = net.ops[0];
        if (a.in_buf == net.in_buf && a.cin == 8 && (a.real_ch & 0xffff) == 3 && a.cout == 64 && a.ksize == 3 && a.stride == 1 &&
            a.act == FRP_ACT_PRELU && a.res_buf < 0 && a.flags == 0 && a.slope_off >= 0 && !getenv("FRP_NO_EMB_STEM")) {
            EmbStemParams ep{};
            ep.x = (const _Float16*)net.bufs[a.in_buf].p;
            ep.M = batch; ep.H = H; ep.W = W;
            ep.w = (const _Float16*)(wbase + a.w_off);
            ep.bias = (const float*)(wbase + a.bias_off);
            ep.slope = (const float*)(wbase + a.slope_off);
            ep.out = (_Float16*)net.bufs[a.out_buf].p;
            ep.n_dev = n_dev;
            // ... and where the conv behind it runs on the 64 -> 64 kernel (conv3x3_c64.hip), that launch computes the stem of its own
            // input patch from the chips: the 64-channel map is written once (the block's shortcut reads it) and never read back by
            // the conv; one launch fewer.  FRP_NO_STEM_FUSE=1: the two launches (A/B runs; the results are the same bits)
            // (the next op that launches: the block's shortcut conv in between rides in a later k-loop - kc_skip - and reads the map then)
            size_t nb = 1;
            while (nb < net.ops.size() && nb < net.kc_skip.size() && net.kc_skip[nb]) ++nb;
            if (nb < net.ops.size() && !getenv("FRP_NO_STEM_FUSE") && small_m <= 0) {
                const frp_conv_op& b = net.ops[nb];
                const bool plain = !(b.flags & ~FRP_FLAG_BORDER_BIAS) && (b.flags & FRP_FLAG_BORDER_BIAS) && b.out2_buf < 0 && b.res_buf < 0;
                const bool chained = !(nb < net.kc_src.size() && net.kc_src[nb] >= 0);
                // (the fused launch reads the chips while it writes both maps: none of the three buffers may be another's alias - this
                // packer pins network inputs, a foreign blob's plan might not)
                fuse_stem = plain && chained && b.in_buf == a.out_buf && b.out_buf != a.out_buf && b.out_buf != a.in_buf && a.out_buf != a.in_buf && b.cin == 64 && b.cout == 64 && b.ksize == 3 &&
                            b.stride == 1 && b.act == FRP_ACT_PRELU && b.slope_off >= 0 && conv3x3_c64_fuses_stem(batch, H, W, h->n_cu);
                fuse_op = nb;
                // who else reads the stem's map?  Only shortcut convs (1x1, stride 2) that ride in a later k-loop: then a quarter of its
                // pixels is all that has to reach HBM
                fuse_even_only = true;
                for (size_t j = 1; j < net.ops.size(); ++j) {
                    const frp_conv_op& o = net.ops[j];
                    if (j == nb || (o.in_buf != a.out_buf && o.res_buf != a.out_buf)) continue;
                    const bool shortcut = j < net.kc_skip.size() && net.kc_skip[j] && o.in_buf == a.out_buf && o.res_buf != a.out_buf &&
                                          o.ksize == 1 && o.stride == 2;
                    if (!shortcut) fuse_even_only = false;
                }
                if (getenv("FRP_STEM_FULL_MAP")) fuse_even_only = false;          // (A/B runs)
            }
            if (fuse_stem) {
                fused = ep;
            } else {
                hipError_t e = launch_emb_stem(ep, h->stream);
                if (e != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("launch_emb_stem: ") + hipGetErrorString(e));
                *launches += 1;
            }
            d[a.out_buf] = {H, W, 64, false};
            *flops += 2.0 * batch * H * W * 9.0 * 3 * 64;
            skip = 1;
            first = false;
        }
    }
    for (const frp_conv_op& op : net.ops) {
        if (&net == &h->det && h->det_op_limit >= 0 && (int)(&op - net.ops.data()) >= h->det_op_limit) break;   // (diagnostic prefix run)
        if (skip) { --skip; continue; }
        if (first && stem) {
            first = false;
            StemParams sp = *stem;
            sp.w = (const _Float16*)(wbase + op.w_off);
            sp.bias = (const float*)(wbase + op.bias_off);
            sp.out = (_Float16*)net.bufs[op.out_buf].p;
            hipError_t e = launch_stem_u8(sp, h->stream);
            if (e != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("launch_stem_u8: ") + hipGetErrorString(e));
            d[op.out_buf] = {sp.Ho, sp.Wo, 32, false};
            *flops += 2.0 * batch * sp.Ho * sp.Wo * 9.0 * 3 * 32;
            *launches += 1;
            continue;
        }
        first = false;
        TensorDims in = d[op.in_buf];
        if (op.flags & FRP_FLAG_FLATTEN) in = {1, 1, in.h * in.w * in.c, false};
        const size_t opi = (size_t)(&op - net.ops.data());
        if (opi < net.kc_skip.size() && net.kc_skip[opi]) {        // a shortcut conv its consumer computes (K-concat): FLOPs charged here
            TensorDims o;
            o.h = (in.h - 1) / op.stride + 1;
            o.w = (in.w - 1) / op.stride + 1;
            o.c = op.cout;
            d[op.out_buf] = o;
            *flops += 2.0 * batch * o.h * o.w * (double)op.cin * op.cout;
            continue;
        }
        ConvParams p{};
        p.x = (const _Float16*)net.bufs[op.in_buf].p;
        p.w = (const _Float16*)(wbase + op.w_off);
        p.bias = (const float*)(wbase + op.bias_off);
        p.slope = op.slope_off >= 0 ? (const float*)(wbase + op.slope_off) : nullptr;
        p.res = op.res_buf >= 0 ? (const _Float16*)net.bufs[op.res_buf].p : nullptr;
        p.out = net.bufs[op.out_buf].p;
        p.N = batch; p.H = in.h; p.W = in.w; p.Cin = op.cin; p.Cout = op.cout;
        p.KS = op.ksize; p.stride = op.stride; p.act = op.act;
        p.n_dev = n_dev;
        p.n_cu = h->n_cu;
        p.small_m = small_m;
        if (s2_optin) p.dbg |= 2048;
        if (fuse_stem && opi == fuse_op) {
            p.stem_x = fused.x; p.stem_w = fused.w; p.stem_bias = fused.bias; p.stem_slope = fused.slope; p.stem_out = fused.out;
            p.stem_even_only = fuse_even_only ? 1 : 0;
        }
        p.wino_wide_only = &net == &h->det ? 1 : 0;
        {
            const size_t oi = (size_t)(&op - net.ops.data());
            if (allow_wino && oi < net.wino_off.size() && net.wino_off[oi] >= 0) p.wino_w = (const _Float16*)(wbase + net.wino_off[oi]);
        }
        p.flags = op.flags & (FRP_FLAG_BORDER_BIAS | FRP_FLAG_OUT_F32 | FRP_FLAG_RES_UP2);
        if (op.flags & FRP_FLAG_RES_UP2) { p.Hr = d[op.res_buf].h; p.Wr = d[op.res_buf].w; }
        p.in_scale = p.out_scale = 1.0f;
        if (op.flags & FRP_OPFLAG_FP8_MFMA) {      // fp8 operands: E4M3 weights as stored, per-cout scales behind them
            const size_t welems = (size_t)op.cout * op.ksize * op.ksize * op.cin;
            p.flags |= FRP_FLAG_F8;
            p.wscale = (const float*)(wbase + op.w_off + (welems + 15) / 16 * 16);
            p.in_scale = op.in_scale;
        }
        if (opi < net.kc_src.size() && net.kc_src[opi] >= 0) {      // K-concat: the block's shortcut conv rides in this conv's k-loop
            const frp_conv_op& sc = net.ops[net.kc_src[opi]];
            p.x2 = (const _Float16*)net.bufs[sc.in_buf].p;
            p.Cin2 = sc.cin;
            p.w = (const _Float16*)(wbase + net.kc_w_off[opi]);
            p.bias = (const float*)(wbase + net.kc_bias_off[opi]);
            p.res = nullptr;
        }
        if (op.flags & FRP_OPFLAG_OUT_FP8) p.flags |= FRP_FLAG_OUT_FP8;
        if (op.out2_buf >= 0) p.out2 = net.bufs[op.out2_buf].p;
        if ((op.flags & FRP_OPFLAG_OUT_FP8) || op.out2_buf >= 0) p.out_scale = op.out_scale;
        // skinny fp32-output GEMM (the FC): split K over the CUs; the slabs are reduced (+bias) by
        // the l2norm kernel that follows
        h->fc_ksplit = 0;
        if ((op.flags & FRP_FLAG_OUT_F32) && &op == &net.ops.back() && !getenv("FRP_NO_SPLITK")) {
            const int ncu = h->n_cu;
            const int hw_out = ((in.h + 2 * (op.ksize / 2) - op.ksize) / op.stride + 1) * ((in.w + 2 * (op.ksize / 2) - op.ksize) / op.stride + 1);
            // device-side count: the kernel picks the factor of the real batch itself (the same function), the slabs are
            // sized for the largest one - that of a single image - times the capacity
            const int ks = conv_pick_ksplit((n_dev ? 1 : batch) * hw_out, op.cout, op.ksize * op.ksize * op.cin, op.flags, op.res_buf >= 0, ncu);
            if (ks > 1) {
                const size_t slab = (size_t)ks * batch * op.cout * 4;   // 1x1 output per image for the FC shape
                if (in.h == 1 && in.w == 1 && ensure(h, h->splitk_ws, slab) == FRP_OK) {
                    p.ksplit = n_dev ? -1 : ks;
                    p.out = h->splitk_ws.p;
                    h->fc_ksplit = p.ksplit;
                    h->fc_bias = p.bias;
                    h->fc_ktot = op.ksize * op.ksize * op.cin;
                }
            }
        }
        // the weights the NEXT launch will stream (a quarter-tile launch with CUs to spare warms the L2s with them: conv_common.h)
        for (size_t nx = opi + 1; nx < net.ops.size(); ++nx) {
            if (nx < net.kc_skip.size() && net.kc_skip[nx]) continue;
            const frp_conv_op& no = net.ops[nx];
            const size_t es = (no.flags & FRP_OPFLAG_FP8_MFMA) ? 1 : 2;
            if (nx < net.kc_src.size() && net.kc_src[nx] >= 0) {
                p.pf_ptr = wbase + net.kc_w_off[nx];
                p.pf_bytes = (unsigned)((size_t)no.cout * (9 * (size_t)no.cin + net.ops[net.kc_src[nx]].cin) * 2);
            } else if (allow_wino && nx < net.wino_off.size() && net.wino_off[nx] >= 0) {
                p.pf_ptr = wbase + net.wino_off[nx];
                p.pf_bytes = (unsigned)std::min<size_t>(conv3x3_wino_image_bytes(no.cin, no.cout), 0x7fffffffu);
            } else {
                p.pf_ptr = wbase + no.w_off;
                p.pf_bytes = (unsigned)std::min<size_t>((size_t)no.cout * no.ksize * no.ksize * no.cin * es, 0x7fffffffu);
            }
            break;
        }
        hipError_t e = launch_conv(p, h->stream);
        if (e != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("launch_conv: ") + hipGetErrorString(e));
        const int pad = op.ksize / 2;
        TensorDims out;
        out.h = (in.h + 2 * pad - op.ksize) / op.stride + 1;
        out.w = (in.w + 2 * pad - op.ksize) / op.stride + 1;
        out.c = op.cout;
        out.f32 = (op.flags & FRP_FLAG_OUT_F32) != 0;
        out.f8 = (op.flags & FRP_OPFLAG_OUT_FP8) != 0;
        d[op.out_buf] = out;
        if (op.out2_buf >= 0) { TensorDims o2 = out; o2.f8 = true; d[op.out2_buf] = o2; }
        if (&net == &h->det && h->det_hash_on && opi < 64 && !n_dev) {      // (diagnostic: hash of this op's output, in stream order)
            e = launch_tensor_hash(net.bufs[op.out_buf].p, (size_t)batch * out.h * out.w * out.c * (out.f32 ? 4 : 2),
                                   (unsigned long long*)h->det_hashes.p + opi, h->stream);
            if (e != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("tensor_hash: ") + hipGetErrorString(e));
        }
        const int cin_r = op.real_ch & 0xffff, cout_r = (op.real_ch >> 16) & 0xffff;
        const double fl = 2.0 * batch * out.h * out.w * (double)op.ksize * op.ksize * (cin_r ? cin_r : op.cin) * (cout_r ? cout_r : op.cout);
        *flops += fl;
        if (op.flags & FRP_OPFLAG_FP8_MFMA) { h->ctr.f8_conv_flops += fl; h->ctr.f8_conv_launches += 1; }
        *launches += 1;
    }
    net.dims = d;
    return FRP_OK;
}

static void drop_graphs(frp_handle* h) {
    for (NetGraph& g : h->graphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    h->graphs.clear();
    h->graph_seen.clear();
    h->graph_bad.clear();
}

// The pass, replayed from its captured graph when it has been asked for before (FRP_NO_GRAPH=1: always launch by launch).  The key names
// everything the launches depend on that is not fixed by the loaded weights: program, shapes, family, operand pointers, every switch
// run_net_body reads; device buffers and weights are covered by the allocation epoch.  Not with stage timers (events between the
// passes are fine, but the diagnostics inside a pass are not captured), not with the detector diagnostics.
int run_net(frp_handle* h, Net& net, int batch, int H, int W, double* flops, int64_t* launches, const StemParams* stem = nullptr,
            const int32_t* n_dev = nullptr, bool allow_wino = true) {
    static const bool off = getenv("FRP_NO_GRAPH") != nullptr;
    if (off || h->det_hash_on || h->det_op_limit >= 0) return run_net_body(h, net, batch, H, W, flops, launches, stem, n_dev, allow_wino);
    char kb[512];
    auto env = [](const char* n) { const char* v = getenv(n); return v ? (v[0] ? v[0] : '1') : '-'; };
    int len = snprintf(kb, sizeof kb, "%c|%d|%d|%d|%d|%p|%p|%c%c%c%c%c%c%c%c|", &net == &h->det ? 'd' : 'e', batch, H, W, (int)allow_wino, (const void*)n_dev,
                       (const void*)h->wdata.p, env("FRP_SMALL_M"), env("FRP_S2"), env("FRP_NO_STEM_FUSE"), env("FRP_STEM_FULL_MAP"),
                       env("FRP_NO_FUSED_STEM12"), env("FRP_NO_EMB_STEM"), env("FRP_NO_SPLITK"), env("FRP_NO_PREFETCH"));
    if (stem && len > 0 && len < (int)sizeof kb)
        len += snprintf(kb + len, sizeof kb - len, "%p|%d|%d|%d|%ld|%ld|%d|%d|%d", (const void*)stem->frames, stem->B, stem->H, stem->W, stem->row_stride,
                        stem->frame_stride, stem->Hc, stem->Wc, stem->rgb_in);
    if (len <= 0 || len >= (int)sizeof kb) return run_net_body(h, net, batch, H, W, flops, launches, stem, n_dev, allow_wino);
    const std::string key(kb);
    for (size_t i = 0; i < h->graphs.size(); ++i) {
        NetGraph& g = h->graphs[i];
        if (g.key != key) continue;
        if (g.epoch != h->alloc_epoch) {            // its buffers may have moved
            (void)hipGraphExecDestroy(g.exec);
            h->graphs.erase(h->graphs.begin() + i);
            break;
        }
        if (hipGraphLaunch(g.exec, h->stream) != hipSuccess) {      // (never seen; if the runtime refuses a replay, the pass is launched instead)
            (void)hipGetLastError();
            (void)hipGraphExecDestroy(g.exec);
            h->graphs.erase(h->graphs.begin() + i);
            h->graph_bad.push_back(key);
            break;
        }
        *flops += g.dflops;
        *launches += g.dlaunches;
        h->ctr.f8_conv_flops += g.df8flops;
        h->ctr.f8_conv_launches += g.df8launches;
        h->fc_ksplit = g.fc_ksplit; h->fc_bias = g.fc_bias; h->fc_ktot = g.fc_ktot;
        net.dims = g.dims;
        h->graph_replays += 1;
        return FRP_OK;
    }
    auto has = [](const std::vector<std::string>& v, const std::string& k) { for (const std::string& x : v) if (x == k) return true; return false; };
    if (has(h->graph_bad, key) || !has(h->graph_seen, key)) {
        if (h->graph_seen.size() > 256) h->graph_seen.clear();
        if (!has(h->graph_seen, key)) h->graph_seen.push_back(key);
        return run_net_body(h, net, batch, H, W, flops, launches, stem, n_dev, allow_wino);
    }
    // second request for this pass: capture it (thread-local mode: the other lanes' threads keep allocating and synchronising as they like)
    const uint64_t epoch0 = h->alloc_epoch;
    const double f0 = *flops, f80 = h->ctr.f8_conv_flops;
    const int64_t l0 = *launches, l80 = h->ctr.f8_conv_launches;
    if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        h->graph_bad.push_back(key);
        return run_net_body(h, net, batch, H, W, flops, launches, stem, n_dev, allow_wino);
    }
    const int rc = run_net_body(h, net, batch, H, W, flops, launches, stem, n_dev, allow_wino);
    hipGraph_t graph = nullptr;
    const hipError_t ce = hipStreamEndCapture(h->stream, &graph);
    NetGraph g;
    bool ok = rc == FRP_OK && ce == hipSuccess && graph && h->alloc_epoch == epoch0;
    if (ok) ok = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0) == hipSuccess;
    if (graph) (void)hipGraphDestroy(graph);
    if (!ok) {
        // nothing of the captured pass has run: say so once, then do it launch by launch (the counters were charged by the capture pass)
        (void)hipGetLastError();
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        h->graph_bad.push_back(key);
        if (rc != FRP_OK) return rc;
        *flops = f0; *launches = l0; h->ctr.f8_conv_flops = f80; h->ctr.f8_conv_launches = l80;
        return run_net_body(h, net, batch, H, W, flops, launches, stem, n_dev, allow_wino);
    }
    g.key = key;
    g.epoch = epoch0;
    g.dflops = *flops - f0; g.dlaunches = *launches - l0;
    g.df8flops = h->ctr.f8_conv_flops - f80; g.df8launches = h->ctr.f8_conv_launches - l80;
    g.fc_ksplit = h->fc_ksplit; g.fc_bias = h->fc_bias; g.fc_ktot = h->fc_ktot;
    g.dims = net.dims;
    if (h->graphs.size() >= 16) {                  // (two frame buffers x two networks x a few call shapes; the oldest goes)
        (void)hipGraphExecDestroy(h->graphs.front().exec);
        h->graphs.erase(h->graphs.begin());
    }
    const hipError_t le = hipGraphLaunch(g.exec, h->stream);
    h->graphs.push_back(g);
    if (le != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("hipGraphLaunch: ") + hipGetErrorString(le));
    return FRP_OK;
}

int parse_net(frp_handle* h, const unsigned char* blob, size_t bytes, uint64_t off, uint32_t n_ops, uint32_t n_bufs,
              uint32_t in_buf, uint32_t in_ch, uint64_t data_bytes, Net& net) {
    if (off > bytes || n_ops > (bytes - off) / sizeof(frp_conv_op)) return fail(h, FRP_ERR_BLOB, "op table out of range");
    if (n_bufs == 0 || n_bufs > 4096 || in_buf >= n_bufs) return fail(h, FRP_ERR_BLOB, "bad buffer count");
    for (DevBuf& b : net.bufs) release(b);
    net.ops.resize(n_ops);
    if (n_ops) memcpy(net.ops.data(), blob + off, (size_t)n_ops * sizeof(frp_conv_op));
    net.n_bufs = (int)n_bufs;
    net.in_buf = (int)in_buf;
    net.in_ch = (int)in_ch;
    net.bufs.assign(n_bufs, DevBuf());
    for (const frp_conv_op& op : net.ops) {
        if (op.in_buf < 0 || op.in_buf >= (int)n_bufs || op.out_buf < 0 || op.out_buf >= (int)n_bufs ||
            op.res_buf >= (int)n_bufs || op.res_buf < -1 || op.in_buf == op.out_buf || op.res_buf == op.out_buf)
            return fail(h, FRP_ERR_BLOB, "op buffer id out of range");
        if ((op.flags & FRP_FLAG_RES_UP2) && op.res_buf < 0) return fail(h, FRP_ERR_BLOB, "upsampled residual without a residual buffer");
        if (op.out2_buf < -1 || op.out2_buf >= (int)n_bufs || op.out2_buf == op.in_buf || op.out2_buf == op.out_buf ||
            (op.out2_buf >= 0 && op.out2_buf == op.res_buf))
            return fail(h, FRP_ERR_BLOB, "op second-output buffer id out of range");
        if ((op.flags & FRP_OPFLAG_FP8_MFMA) && !(op.flags & FRP_OPFLAG_W_FP8)) return fail(h, FRP_ERR_BLOB, "fp8 op without fp8 weights");
        if ((op.flags & (FRP_OPFLAG_FP8_MFMA | FRP_OPFLAG_OUT_FP8)) || op.out2_buf >= 0) {
            if (!(op.in_scale > 0.f) || !(op.out_scale > 0.f) || !std::isfinite(op.in_scale) || !std::isfinite(op.out_scale))
                return fail(h, FRP_ERR_BLOB, "fp8 tensor scale must be positive and finite");
        }
        if (!(op.ksize == 1 || op.ksize == 3) || !(op.stride == 1 || op.stride == 2) || op.cin < 8 || (op.cin & 7) ||
            op.cout < 4 || (op.cout & 3) || op.act < 0 || op.act > 2)
            return fail(h, FRP_ERR_BLOB, "op shape not supported");
        const uint64_t welems = (uint64_t)op.cout * op.ksize * op.ksize * op.cin;
        // fp8 storage: one byte per element, then (16-byte aligned) cout fp32 scales
        const uint64_t wbytes = (op.flags & FRP_OPFLAG_W_FP8) ? ((welems + 15) / 16 * 16 + (uint64_t)op.cout * 4) : welems * 2;
        const uint64_t bbytes = (uint64_t)op.cout * 4 * ((op.flags & FRP_FLAG_BORDER_BIAS) ? 9 : 1);
        if (op.w_off < 0 || (uint64_t)op.w_off + wbytes > data_bytes || (op.w_off & 15) || op.bias_off < 0 ||
            (uint64_t)op.bias_off + bbytes > data_bytes || (op.bias_off & 15))
            return fail(h, FRP_ERR_BLOB, "op tensor offset out of range");
        if (op.act == FRP_ACT_PRELU &&
            (op.slope_off < 0 || (uint64_t)op.slope_off + (uint64_t)op.cout * 4 > data_bytes || (op.slope_off & 15)))
            return fail(h, FRP_ERR_BLOB, "op slope offset out of range");
    }
    return FRP_OK;
}

int upload_frames(frp_handle* h, const uint8_t* bgr, int B, int H, int W, int64_t row_stride) {
    if (!bgr || B <= 0 || H <= 0 || W <= 0 || row_stride < (int64_t)W * 3) return fail(h, FRP_ERR_INVALID, "bad frame arguments");
    if (B > 1024) return fail(h, FRP_ERR_INVALID, "batch too large (max 1024 frames per call)");
    FRPCHK(ensure(h, h->frames, (size_t)B * H * W * 3));
    rec(h, EV_START);
    HIPCHK(h, hipMemcpy2DAsync(h->frames.p, (size_t)W * 3, bgr, (size_t)row_stride, (size_t)W * 3, (size_t)B * H,
                               hipMemcpyHostToDevice, h->stream));
    rec(h, EV_H2D);
    h->rB = B; h->rH = H; h->rW = W;
    h->dH = H; h->dW = W; h->det_scaled = false;
    h->canvas_h = round_up(H, 32);
    h->canvas_w = round_up(W, 32);
    return FRP_OK;
}

// choose the detector source: the resident frames (Hs,Ws == frame size) or a bilinear resize of them
int select_det_source(frp_handle* h, int Hs, int Ws) {
    if (h->rB <= 0) return fail(h, FRP_ERR_INVALID, "no resident frames (call frp_upload_frames)");
    if (Hs <= 0 || Ws <= 0 || Hs > 16384 || Ws > 16384) return fail(h, FRP_ERR_INVALID, "bad detector size");
    if (Hs == h->rH && Ws == h->rW) {
        h->det_scaled = false;
    } else {
        FRPCHK(ensure(h, h->scaled, (size_t)h->rB * Hs * Ws * 3));
        hipError_t e = launch_resize_u8((const uint8_t*)h->frames.p, h->rB, h->rH, h->rW, (uint8_t*)h->scaled.p, Hs, Ws, h->stream);
        if (e != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("resize: ") + hipGetErrorString(e));
        h->det_scaled = true;
    }
    h->dH = Hs; h->dW = Ws;
    h->canvas_h = round_up(Hs, 32);
    h->canvas_w = round_up(Ws, 32);
    return FRP_OK;
}

int ensure_results(frp_handle* h, int B, int K) {
    const size_t s = (size_t)B * K;
    FRPCHK(ensure(h, h->boxes, s * 4 * 4));
    FRPCHK(ensure(h, h->kps, s * 10 * 4));
    FRPCHK(ensure(h, h->scores, s * 4));
    FRPCHK(ensure(h, h->anchor, s * 4));
    FRPCHK(ensure(h, h->counts, (size_t)B * 4));
    FRPCHK(ensure(h, h->face_slot, s * 4));
    FRPCHK(ensure(h, h->nfaces, 16));
    return FRP_OK;
}

int run_detect(frp_handle* h, int K, float det_thresh, float nms_iou, uint32_t flags) {
    if (!h->have_weights) return fail(h, FRP_ERR_NO_WEIGHTS, "no weights loaded");
    if (h->rB <= 0) return fail(h, FRP_ERR_INVALID, "no resident frames (call frp_upload_frames)");
    if (K <= 0 || K > FRP_MAX_FACES_CAP) return fail(h, FRP_ERR_INVALID, "max_faces out of range");
    const int B = h->rB, Hc = h->canvas_h, Wc = h->canvas_w;
    // The detector's first layer reads the u8 frames directly (fused normalise + conv) whenever
    // the program starts with the standard 3x3 s2 3->32 stem; FRP_NO_FUSED_STEM=1 keeps the
    // two-kernel path (preprocess to an NHWC8 blob, then the generic conv) for A/B runs.
    const bool fused = stem_fusable(h->det) && !getenv("FRP_NO_FUSED_STEM");
    FRPCHK(plan_net(h, h->det, B, Hc, Wc, fused));
    FRPCHK(ensure_results(h, B, K));
    hipError_t e = hipSuccess;
    StemParams sp{};
    const uint8_t* dsrc = h->det_scaled ? (const uint8_t*)h->scaled.p : (const uint8_t*)h->frames.p;
    if (fused) {
        sp.frames = dsrc;
        sp.B = B; sp.H = h->dH; sp.W = h->dW;
        sp.row_stride = (long)h->dW * 3; sp.frame_stride = (long)h->dH * h->dW * 3;
        sp.Hc = Hc; sp.Wc = Wc; sp.Ho = Hc / 2; sp.Wo = Wc / 2;
        sp.rgb_in = (flags & FRP_FLAG_RGB) ? 1 : 0;
    } else {
        e = launch_preprocess(dsrc, B, h->dH, h->dW, (long)h->dW * 3, (long)h->dH * h->dW * 3,
                              (_Float16*)h->det.bufs[h->det.in_buf].p, Hc, Wc, (flags & FRP_FLAG_RGB) ? 1 : 0, h->stream);
        if (e != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("preprocess: ") + hipGetErrorString(e));
    }
    rec(h, EV_PRE);
    // Winograd family for the detector's wide 128 / 256-channel layers: from two rounds of 8 x 30 tiles on its stride-8 maps on
    // (1080p: four frames; below that the direct family with its quarter tiles and weight prefetch: single-image latency)
    const bool det_wino = (long)B * (Hc / 8) * (Wc / 8) >= 2L * 240 * (h->n_cu > 0 ? h->n_cu : 256);
    FRPCHK(run_net(h, h->det, B, Hc, Wc, &h->ctr.det_conv_flops, &h->ctr.det_conv_launches, fused ? &sp : nullptr, nullptr, det_wino));
    rec(h, EV_DET);
    DecodeParams dp{};
    for (int l = 0; l < 3; ++l) {
        const int bi = (int)h->hdr.det_head_buf[l];
        dp.head[l] = (const _Float16*)h->det.bufs[bi].p;
        dp.hl[l] = h->det.dims[bi].h;
        dp.wl[l] = h->det.dims[bi].w;
        if (h->det.dims[bi].c != 32) return fail(h, FRP_ERR_BLOB, "detector head must have 32 channels");
    }
    dp.B = B;
    dp.max_faces = K;
    const bool forced = flags & FRP_FLAG_FORCED_K;
    dp.logit_thresh = forced ? -INFINITY : logit_threshold(det_thresh);
    dp.nms_iou = forced ? 2.0f : nms_iou;
    dp.boxes = (float*)h->boxes.p; dp.kps = (float*)h->kps.p; dp.scores = (float*)h->scores.p;
    dp.anchor = (int32_t*)h->anchor.p; dp.counts = (int32_t*)h->counts.p;
    {
        size_t anchors = 0;
        for (int l = 0; l < 3; ++l) anchors += (size_t)dp.hl[l] * dp.wl[l] * 2;
        FRPCHK(ensure(h, h->dense_logits, (size_t)B * anchors * 2));
        dp.logits = (_Float16*)h->dense_logits.p;
    }
    e = launch_decode_nms(dp, h->stream);
    if (e != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("decode_nms: ") + hipGetErrorString(e));
    e = launch_compact_faces((const int32_t*)h->counts.p, B, K, (int32_t*)h->face_slot.p, (int32_t*)h->nfaces.p, h->stream);
    if (e != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("compact_faces: ") + hipGetErrorString(e));
    rec(h, EV_DEC);
    h->last_B = B;
    h->last_K = K;
    return FRP_OK;
}

// chips for n faces are in emb.bufs[in]; run embedder + l2norm (+ fp16 copy for the matcher)
// Kernel family of an embedder pass.  The Winograd kernel (256 x 128 tiles only) pays from about 128 faces up (end to end,
// tools/family_crossover.py: 80 slots 5.06 vs 4.62 ms, 100 / 128 slots equal, 160 slots 7.58 vs 7.78); below, the direct kernels in
// quarter tiles are up to 2.6 x faster per layer (tools/small_m_probe.py).  The two families differ in the last
// bits (1 - cos 1.6e-6), so the choice is made ONCE per call, from a count that does not depend on what the detector
// found: `family_count` = the slots of the call (B x K of a process call, whether the face count stays on the device or
// not; the faces handed to the embed / finish calls).  Within a family every tile size gives the same bits, so a face's
// embedding depends on the call's slot count being above or below FRP_WINO_MIN_FACES and on nothing else in the batch.
#define FRP_WINO_MIN_FACES 128
int run_embed(frp_handle* h, int n, const int32_t* n_dev = nullptr, int family_count = -1) {
    if (n <= 0) return FRP_OK;
    if (family_count < 0) family_count = n;
    const char* wm = getenv("FRP_WINO_MIN_FACES");               // A/B runs and tests that pin the family
    const int wino_min = wm ? atoi(wm) : FRP_WINO_MIN_FACES;
    FRPCHK(run_net(h, h->emb, n, FRP_CHIP, FRP_CHIP, &h->ctr.emb_conv_flops, &h->ctr.emb_conv_launches, nullptr, n_dev,
                   family_count >= wino_min));
    rec(h, EV_EMB);
    const int mpad = round_up(n, 32);
    FRPCHK(ensure(h, h->q16, (size_t)mpad * FRP_EMB_DIM * 2));
    HIPCHK(h, hipMemsetAsync(h->q16.p, 0, (size_t)mpad * FRP_EMB_DIM * 2, h->stream));
    hipError_t e = launch_l2norm((float*)h->emb.bufs[h->hdr.emb_out_buf].p, (_Float16*)h->q16.p, n, FRP_EMB_DIM, h->stream,
                                 h->fc_ksplit != 0 && h->fc_ksplit != 1 ? (const float*)h->splitk_ws.p : nullptr, h->fc_ksplit, h->fc_bias,
                                 n_dev, h->fc_ktot, h->n_cu);
    if (e != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("l2norm: ") + hipGetErrorString(e));
    rec(h, EV_L2);
    return FRP_OK;
}

// q16 [mpad,512] holds n unit queries -> best_idx/best_cos [n]
int run_match(frp_handle* h, int n, float* all_scores_dev, const int32_t* n_dev = nullptr) {
    if (n <= 0) return FRP_OK;
    if (h->g_rows <= 0) return fail(h, FRP_ERR_NO_GALLERY, "gallery is empty");
    const int mpad = round_up(n, 32);
    MatchParams mp{};
    mp.gallery = (const _Float16*)h->gallery.p;
    mp.N = h->g_rows;
    mp.q = (const _Float16*)h->q16.p;
    mp.M = n;
    mp.Mpad = mpad;
    mp.n_wg = match_num_workgroups(h->g_rows);
    FRPCHK(ensure(h, h->part_cos, (size_t)mp.n_wg * mpad * 4));
    FRPCHK(ensure(h, h->part_idx, (size_t)mp.n_wg * mpad * 4));
    FRPCHK(ensure(h, h->best_cos, (size_t)mpad * 4));
    FRPCHK(ensure(h, h->best_idx, (size_t)mpad * 4));
    mp.part_cos = (float*)h->part_cos.p; mp.part_idx = (int32_t*)h->part_idx.p;
    mp.best_cos = (float*)h->best_cos.p; mp.best_idx = (int32_t*)h->best_idx.p;
    mp.all_scores = all_scores_dev;
    mp.n_dev = n_dev;
    hipError_t e = launch_match(mp, h->stream);
    if (e != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("match: ") + hipGetErrorString(e));
    h->ctr.match_bytes += (double)h->g_rows * FRP_EMB_DIM * 2;
    h->ctr.match_launches += 1;
    return FRP_OK;
}

// align + embed + match for the faces listed in h->kps / h->counts / h->face_slot (device), always from
// the full-resolution resident frames.  n_known >= 0: face count known on the host.
int resolve_count(frp_handle* h);
int run_faces(frp_handle* h, int K, int n_known, uint32_t flags) {
    const int B = h->rB;
    int n;
    // a device-count pass that nobody fetched or synchronised yet (two process calls queued back to back): its count and its
    // counter corrections live in single slots (h_nfaces, pend_*) this pass is about to reuse - settle it first
    if (h->last_nfaces < 0) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        FRPCHK(resolve_count(h));
    }
    // Threshold mode (the reference's loop, routes/camera.py:232-259): how many faces the detector kept is known on the
    // device only.  It STAYS there: align, the embedder's kernels, the l2norm and the matcher are launched for the capacity
    // B x K and read the count from device memory (grids sized for the capacity; workgroups beyond the real tiles leave at
    // once), so the pipeline has no host round trip.  The host learns the count with the results (frp_fetch_results).
    // FRP_HOST_COUNT=1 keeps the former path (copy the count, wait, launch for exactly n) for A/B runs - both give the
    // same bits.  More than FRP_MATCH_TOP1_MAX slots: the per-tile matcher has no device-count form, former path.
    // (FRP_MATCH_V1=1 pins the per-tile matcher for A/B runs: it has no device-count form either)
    const bool dev_count = n_known < 0 && round_up(B * K, 32) <= FRP_MATCH_TOP1_MAX && !getenv("FRP_HOST_COUNT") && !getenv("FRP_MATCH_V1");
    const int32_t* n_dev = nullptr;
    if (n_known >= 0) {
        n = n_known;
    } else if (dev_count) {
        n = B * K;
        n_dev = (const int32_t*)h->nfaces.p;
        HIPCHK(h, hipMemcpyAsync(h->h_nfaces, h->nfaces.p, 4, hipMemcpyDeviceToHost, h->stream));   // read after the next wait
    } else {
        HIPCHK(h, hipMemcpyAsync(h->h_nfaces, h->nfaces.p, 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        n = *h->h_nfaces;
        if (n < 0 || n > B * K) return fail(h, FRP_ERR_HIP, "corrupt face count");
    }
    h->last_nfaces = n_dev ? -1 : n;          // -1: pending, resolved by the next call that waits for the stream
    h->last_cap = n;
    h->last_matched = false;
    const double flops0 = h->ctr.emb_conv_flops, f8flops0 = h->ctr.f8_conv_flops;
    if (n > 0) {
        FRPCHK(plan_net(h, h->emb, n, FRP_CHIP, FRP_CHIP));
        AlignParams ap{};
        ap.frames = (const uint8_t*)h->frames.p;
        ap.B = B; ap.H = h->rH; ap.W = h->rW;
        ap.row_stride = (long)h->rW * 3;
        ap.frame_stride = (long)h->rH * h->rW * 3;
        ap.kps = (const float*)h->kps.p;
        ap.counts = (const int32_t*)h->counts.p;
        ap.max_faces = K;
        ap.face_slot = (const int32_t*)h->face_slot.p;
        ap.n_faces = n;
        ap.n_dev = n_dev;
        ap.rgb_in = (flags & FRP_FLAG_RGB) ? 1 : 0;
        ap.chips = (_Float16*)h->emb.bufs[h->emb.in_buf].p;
        hipError_t e = launch_align(ap, h->stream);
        if (e != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("align: ") + hipGetErrorString(e));
        rec(h, EV_ALIGN);
        FRPCHK(run_embed(h, n, n_dev, n_known >= 0 ? n_known : B * K));
        if (!(flags & FRP_FLAG_NO_MATCH) && h->g_rows > 0) {
            FRPCHK(run_match(h, n, nullptr, n_dev));
            h->last_matched = true;
        }
        rec(h, EV_MATCH);
    } else {
        rec(h, EV_ALIGN); rec(h, EV_EMB); rec(h, EV_L2); rec(h, EV_MATCH);
    }
    h->ctr.frames += B;
    if (n_dev) {
        h->pend_flops = h->ctr.emb_conv_flops - flops0;      // charged for the capacity: corrected once the count is known
        h->pend_f8flops = h->ctr.f8_conv_flops - f8flops0;
        h->pend_cap = n;
    } else {
        h->ctr.faces += n;
    }
    return FRP_OK;
}

// the face count of a device-count pass, once the stream has been waited for (h_nfaces was copied behind the pass)
int resolve_count(frp_handle* h) {
    if (h->last_nfaces >= 0) return FRP_OK;
    int n = *h->h_nfaces;
    const bool corrupt = n < 0 || n > h->last_cap;     // (the host-count path fails the same way: "corrupt face count")
    if (corrupt) n = 0;
    h->last_nfaces = n;
    h->ctr.faces += n;
    if (h->pend_cap > 0) {
        h->ctr.emb_conv_flops += h->pend_flops * n / h->pend_cap - h->pend_flops;
        h->ctr.f8_conv_flops += h->pend_f8flops * n / h->pend_cap - h->pend_f8flops;
    }
    h->pend_cap = 0;
    h->pend_flops = h->pend_f8flops = 0.0;
    return corrupt ? fail(h, FRP_ERR_HIP, "corrupt face count") : FRP_OK;
}

int run_pipeline(frp_handle* h, int K, float det_thresh, float nms_iou, uint32_t flags) {
    if (h->det_scaled) FRPCHK(select_det_source(h, h->rH, h->rW));     // the fused path always detects at full size
    FRPCHK(run_detect(h, K, det_thresh, nms_iou, flags));
    const long A = (long)h->det.dims[h->hdr.det_head_buf[0]].h * h->det.dims[h->hdr.det_head_buf[0]].w * 2;
    return run_faces(h, K, ((flags & FRP_FLAG_FORCED_K) && A >= K) ? h->rB * K : -1, flags);   // forced-K: count known
}

void accumulate_events(frp_handle* h, bool with_h2d) {
    if (!h->cfg.profile) return;
    auto el = [&](int a, int b) { float ms = 0.f; return hipEventElapsedTime(&ms, h->ev[a], h->ev[b]) == hipSuccess ? (double)ms : 0.0; };
    frp_counters& c = h->ctr;
    if (with_h2d) c.ms_h2d += el(EV_START, EV_H2D);
    c.ms_preprocess += el(EV_H2D, EV_PRE);
    c.ms_det_conv += el(EV_PRE, EV_DET);
    c.ms_decode += el(EV_DET, EV_DEC);
    c.ms_align += el(EV_DEC, EV_ALIGN);
    c.ms_emb_conv += el(EV_ALIGN, EV_EMB);
    c.ms_l2norm += el(EV_EMB, EV_L2);
    c.ms_match += el(EV_L2, EV_MATCH);
    c.ms_total += el(with_h2d ? EV_START : EV_H2D, EV_MATCH);
}

// the split entry points (pyramid: frp_detect_resident, frp_finish_faces) account their half of the stage times
void accumulate_detect_events(frp_handle* h) {
    if (!h->cfg.profile) return;
    auto el = [&](int a, int b) { float ms = 0.f; return hipEventElapsedTime(&ms, h->ev[a], h->ev[b]) == hipSuccess ? (double)ms : 0.0; };
    h->ctr.ms_preprocess += el(EV_H2D, EV_PRE);          // incl. the pyramid resize
    h->ctr.ms_det_conv += el(EV_PRE, EV_DET);
    h->ctr.ms_decode += el(EV_DET, EV_DEC);
    h->ctr.ms_total += el(EV_H2D, EV_DEC);
}
void accumulate_face_events(frp_handle* h) {
    if (!h->cfg.profile) return;
    auto el = [&](int a, int b) { float ms = 0.f; return hipEventElapsedTime(&ms, h->ev[a], h->ev[b]) == hipSuccess ? (double)ms : 0.0; };
    h->ctr.ms_align += el(EV_DEC, EV_ALIGN);
    h->ctr.ms_emb_conv += el(EV_ALIGN, EV_EMB);
    h->ctr.ms_l2norm += el(EV_EMB, EV_L2);
    h->ctr.ms_match += el(EV_L2, EV_MATCH);
    h->ctr.ms_total += el(EV_DEC, EV_MATCH);
}

// frp_process_resident returns without waiting for the device also when the stage timers are on: its events are read
// by whichever entry point next waits for the stream anyway (frp_fetch_results, frp_synchronize), or - with a wait of
// their own - by the first other call on the handle, before it could re-record them.  (Reading them inside
// frp_process_resident cost a full stream drain per step: 0.8 ms of a 15 ms step in bench.py's fetch-every-step loop.)
void settle_events(frp_handle* h, bool stream_is_idle) {
    if (!h->ev_pending) return;
    h->ev_pending = false;
    if (!stream_is_idle && hipStreamSynchronize(h->stream) != hipSuccess) return;
    accumulate_events(h, false);
}

// page-locked staging for the result fetch, grown on demand.  Device -> PAGEABLE host copies go through the runtime's own
// bounce buffers with whole-device synchronisation semantics: next to torch / RCCL in the process they serialised the
// copy stream's upload of the next batch behind the fetch (the overlapped loop lost its overlap: 20 vs 14.7 ms per
// step); device -> pinned copies are plain stream-ordered DMA on the handle's own stream.
int ensure_pinned(frp_handle* h, size_t bytes) {
    if (bytes <= h->pin_cap && h->pin_stage) return FRP_OK;
    if (h->pin_stage) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        (void)hipHostFree(h->pin_stage);
        h->pin_stage = nullptr;
        h->pin_cap = 0;
    }
    const size_t want = std::max<size_t>(bytes, 1 << 20);
    void* p = nullptr;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) return fail(h, FRP_ERR_OOM, "hipHostMalloc (result staging) failed");
    h->pin_stage = (unsigned char*)p;
    h->pin_cap = want;
    return FRP_OK;
}

int fetch_results(frp_handle* h, float* boxes, float* kps, float* scores, int32_t* counts, float* emb,
                  int32_t* match_idx, float* match_cos) {
    const int B = h->last_B, K = h->last_K;
    if (B <= 0) return fail(h, FRP_ERR_INVALID, "nothing to fetch");
    if (h->last_nfaces < 0) {                  // device-count pass: one short wait for the count, then copy exactly n rows
        HIPCHK(h, hipStreamSynchronize(h->stream));
        FRPCHK(resolve_count(h));
    }
    const int n = h->last_nfaces;
    const size_t s = (size_t)B * K;
    const bool want_emb = n > 0 && emb, want_match = n > 0 && h->last_matched && (match_idx || match_cos);
    // staging layout: counts | boxes | kps | scores | emb (compact, n rows) | idx | cos
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_cnt = take((size_t)B * 4), o_box = take(boxes ? s * 16 : 0), o_kps = take(kps ? s * 40 : 0),
                 o_sc = take(scores ? s * 4 : 0), o_emb = take(want_emb ? (size_t)n * FRP_EMB_DIM * 4 : 0),
                 o_idx = take(want_match ? (size_t)n * 4 : 0), o_cos = take(want_match ? (size_t)n * 4 : 0);
    FRPCHK(ensure_pinned(h, off));
    unsigned char* st = h->pin_stage;
    HIPCHK(h, hipMemcpyAsync(st + o_cnt, h->counts.p, (size_t)B * 4, hipMemcpyDeviceToHost, h->stream));
    if (boxes) HIPCHK(h, hipMemcpyAsync(st + o_box, h->boxes.p, s * 16, hipMemcpyDeviceToHost, h->stream));
    if (kps) HIPCHK(h, hipMemcpyAsync(st + o_kps, h->kps.p, s * 40, hipMemcpyDeviceToHost, h->stream));
    if (scores) HIPCHK(h, hipMemcpyAsync(st + o_sc, h->scores.p, s * 4, hipMemcpyDeviceToHost, h->stream));
    if (want_emb)
        HIPCHK(h, hipMemcpyAsync(st + o_emb, h->emb.bufs[h->hdr.emb_out_buf].p, (size_t)n * FRP_EMB_DIM * 4, hipMemcpyDeviceToHost, h->stream));
    if (want_match) {
        HIPCHK(h, hipMemcpyAsync(st + o_idx, h->best_idx.p, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(st + o_cos, h->best_cos.p, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    settle_events(h, true);
    const int32_t* cnt = (const int32_t*)(st + o_cnt);
    if (counts) memcpy(counts, cnt, (size_t)B * 4);
    if (boxes) memcpy(boxes, st + o_box, s * 16);
    if (kps) memcpy(kps, st + o_kps, s * 40);
    if (scores) memcpy(scores, st + o_sc, s * 4);
    const float* cemb = (const float*)(st + o_emb);
    const int32_t* cidx = (const int32_t*)(st + o_idx);
    const float* ccos = (const float*)(st + o_cos);
    // compact face list -> [B][K] slots; slots beyond counts[b] are zero / -1
    int f = 0;
    for (int b = 0; b < B; ++b) {
        const int nb = std::max(0, std::min(cnt[b], K));
        for (int k = 0; k < K; ++k) {
            const size_t slot = (size_t)b * K + k;
            const bool live = k < nb && f < n;
            if (emb) {
                if (live && want_emb) memcpy(emb + slot * FRP_EMB_DIM, cemb + (size_t)f * FRP_EMB_DIM, FRP_EMB_DIM * 4);
                else memset(emb + slot * FRP_EMB_DIM, 0, FRP_EMB_DIM * 4);
            }
            if (match_idx) match_idx[slot] = (live && want_match) ? cidx[f] : -1;
            if (match_cos) match_cos[slot] = (live && want_match) ? ccos[f] : -1.f;
            if (live) ++f;
        }
    }
    return FRP_OK;
}

int to_f32(frp_handle* h, const void* src, size_t count, int dtype, std::vector<float>& out) {
    out.resize(count);
    if (dtype == FRP_F32) {
        memcpy(out.data(), src, count * 4);
    } else if (dtype == FRP_F64) {
        const double* d = (const double*)src;
        for (size_t i = 0; i < count; ++i) out[i] = (float)d[i];
    } else if (dtype == FRP_F16) {
        const _Float16* d = (const _Float16*)src;
        for (size_t i = 0; i < count; ++i) out[i] = (float)d[i];
    } else {
        return fail(h, FRP_ERR_INVALID, "unknown dtype");
    }
    return FRP_OK;
}

// host fp32 rows -> unit fp16 rows at dst (device), via the scratch buffer
int upload_rows_normalized(frp_handle* h, const float* rows, int64_t n, _Float16* dst) {
    if (n <= 0) return FRP_OK;
    const int64_t chunk = 1 << 16;
    FRPCHK(ensure(h, h->scratch, (size_t)std::min(n, chunk) * FRP_EMB_DIM * 4));
    for (int64_t r0 = 0; r0 < n; r0 += chunk) {
        const int64_t m = std::min(chunk, n - r0);
        HIPCHK(h, hipMemcpyAsync(h->scratch.p, rows + r0 * FRP_EMB_DIM, (size_t)m * FRP_EMB_DIM * 4, hipMemcpyHostToDevice, h->stream));
        hipError_t e = launch_gallery_normalize((const float*)h->scratch.p, dst + r0 * FRP_EMB_DIM, m, FRP_EMB_DIM, h->stream);
        if (e != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("gallery_normalize: ") + hipGetErrorString(e));
        HIPCHK(h, hipStreamSynchronize(h->stream));   // scratch is reused by the next chunk
    }
    return FRP_OK;
}

// exact compat rows: `n` host rows of `d` = 512 values of `dtype` -> float64 at dst (device row pointer)
int upload_rows_exact(frp_handle* h, const void* emb, int64_t n, int dtype, double* dst) {
    if (n <= 0) return FRP_OK;
    const size_t cnt = (size_t)n * FRP_EMB_DIM;
    std::vector<double> tmp;
    const double* src = (const double*)emb;
    if (dtype != FRP_F64) {
        tmp.resize(cnt);
        if (dtype == FRP_F32) { const float* f = (const float*)emb; for (size_t i = 0; i < cnt; ++i) tmp[i] = (double)f[i]; }
        else if (dtype == FRP_F16) { const uint16_t* u = (const uint16_t*)emb; for (size_t i = 0; i < cnt; ++i) tmp[i] = (double)f16_bits_to_f32(u[i]); }
        else return fail(h, FRP_ERR_INVALID, "bad dtype");
        src = tmp.data();
    }
    HIPCHK(h, hipMemcpyAsync(dst, src, cnt * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));      // (tmp / the caller's rows go away)
    return FRP_OK;
}

// a fresh exact matrix of `n` rows widened from unit fp16 device rows (rows installed from device data, or that existed before)
int exact_from_f16(frp_handle* h, const void* dev_f16, int64_t n, DevBuf& fresh) {
    if (n <= 0) return FRP_OK;
    FRPCHK(ensure(h, fresh, (size_t)n * FRP_EMB_DIM * 8));
    hipError_t e = launch_gallery_widen((const _Float16*)dev_f16, (double*)fresh.p, n, FRP_EMB_DIM, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { release(fresh); return fail(h, FRP_ERR_HIP, std::string("gallery_widen: ") + hipGetErrorString(e)); }
    return FRP_OK;
}

struct Guard {
    std::lock_guard<std::mutex> lk;
    explicit Guard(frp_handle* h, bool settle = true) : lk(h->mu) {
        (void)hipSetDevice(h->device);
        if (settle) settle_events(h, false);
    }
};

}  // namespace

extern "C" {

const char* frp_version(void) { return "frp 0.1 (gfx950)"; }

int frp_create(int device, const frp_config* cfg, frp_handle** out) {
    if (!out) return FRP_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FRP_ERR_HIP;   // fail loudly: no CPU fallback
    if (device < 0 || device >= ndev) return FRP_ERR_INVALID;
    frp_handle* h = new (std::nothrow) frp_handle();
    if (!h) return FRP_ERR_OOM;
    h->device = device;
    h->cfg.struct_size = sizeof(frp_config);
    h->cfg.max_batch = 32; h->cfg.max_faces = 10; h->cfg.max_h = 1080; h->cfg.max_w = 1920; h->cfg.profile = 0;
    if (cfg) {
        if (cfg->struct_size != (int32_t)sizeof(frp_config)) { delete h; return FRP_ERR_INVALID; }
        if (cfg->max_batch > 0) h->cfg.max_batch = cfg->max_batch;
        if (cfg->max_faces > 0) h->cfg.max_faces = std::min<int>(cfg->max_faces, FRP_MAX_FACES_CAP);
        if (cfg->max_h > 0) h->cfg.max_h = cfg->max_h;
        if (cfg->max_w > 0) h->cfg.max_w = cfg->max_w;
        h->cfg.profile = cfg->profile ? 1 : 0;
    }
    bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < EV_COUNT; ++i) ok = hipEventCreate(&h->ev[i]) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&h->h_nfaces, 64, hipHostMallocDefault) == hipSuccess;
    if (ok) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) h->n_cu = prop.multiProcessorCount;
    }
    // The copy stream gets its own PRIORITY class: the runtime multiplexes the streams of one class onto a few hardware
    // queues (4 by default), and next to torch's and RCCL's streams in the process the staged upload shared a queue with
    // the compute stream and serialised behind the step's kernels (overlapped loop 18.7-20.6 instead of 14.7 ms per
    // step; GPU_MAX_HW_QUEUES=8 restored it).  A stream of another priority class is not pooled with them.
    if (ok) {
        int lo = 0, hi = 0;
        hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (e == hipSuccess && hi != lo) ok = hipStreamCreateWithPriority(&h->copy_stream, hipStreamNonBlocking, hi) == hipSuccess;
        else ok = hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking) == hipSuccess;
    }
    ok = ok && hipEventCreateWithFlags(&h->ev_next_ready, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&h->ev_next_free, hipEventDisableTiming) == hipSuccess;
    if (!ok) { frp_destroy(h); return FRP_ERR_HIP; }
    h->ctr.struct_size = sizeof(frp_counters);
    *out = h;
    return FRP_OK;
}

void frp_destroy(frp_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
    if (h->comm) { (void)rccl().CommDestroy((ncclComm_t)h->comm); h->comm = nullptr; }
    drop_graphs(h);
    for (DevBuf& b : h->det.bufs) release(b);
    for (DevBuf& b : h->emb.bufs) release(b);
    DevBuf* all[] = {&h->wdata, &h->frames, &h->frames_next, &h->boxes, &h->kps, &h->scores, &h->counts, &h->anchor, &h->face_slot, &h->nfaces,
                     &h->q16, &h->part_cos, &h->part_idx, &h->best_cos, &h->best_idx, &h->scratch, &h->splitk_ws, &h->dense_logits, &h->scaled, &h->gallery,
                     &h->g_reserved, &h->gx, &h->gx_q, &h->gx_out, &h->jpeg_coef, &h->jpeg_planes, &h->jpeg_scan, &h->jpeg_err, &h->det_hashes};
    for (DevBuf* b : all) release(*b);
    for (int i = 0; i < EV_COUNT; ++i) if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
    for (int i = 0; i < 2; ++i) {
        if (h->jpeg_pin[i]) (void)hipHostFree(h->jpeg_pin[i]);
        if (h->ev_jpeg_h2d[i]) (void)hipEventDestroy(h->ev_jpeg_h2d[i]);
    }
    if (h->h_nfaces) (void)hipHostFree(h->h_nfaces);
    if (h->pin_stage) (void)hipHostFree(h->pin_stage);
    for (void* p : h->pinned) (void)hipHostFree(p);
    if (h->ev_next_ready) (void)hipEventDestroy(h->ev_next_ready);
    if (h->ev_next_free) (void)hipEventDestroy(h->ev_next_free);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char* frp_last_error(const frp_handle* h) { return h ? h->err.c_str() : "null handle"; }

int frp_load_weights(frp_handle* h, const void* blob, size_t bytes) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    h->have_weights = false;          // a failed load never leaves a half-replaced program runnable
    ++h->alloc_epoch;                 // (captured passes of the old program are stale even where the new one lands in the same allocation)
    if (!blob || bytes < sizeof(frp_blob_header)) return fail(h, FRP_ERR_BLOB, "blob too small");
    frp_blob_header hd;
    memcpy(&hd, blob, sizeof(hd));
    if (memcmp(hd.magic, FRP_BLOB_MAGIC, 8) != 0 || hd.version != FRP_BLOB_VERSION || hd.header_bytes != sizeof(frp_blob_header))
        return fail(h, FRP_ERR_BLOB, "bad magic/version");
    if (hd.data_offset > bytes || hd.data_bytes > bytes - hd.data_offset) return fail(h, FRP_ERR_BLOB, "data section out of range");
    if (hd.emb_dim != FRP_EMB_DIM || hd.emb_size != FRP_CHIP || hd.det_in_ch != 8 || hd.emb_in_ch != 8 || hd.det_num_anchors != 2)
        return fail(h, FRP_ERR_BLOB, "unsupported network geometry");
    const unsigned char* b = (const unsigned char*)blob;
    FRPCHK(parse_net(h, b, bytes, hd.det_ops_offset, hd.n_det_ops, hd.n_det_bufs, hd.det_in_buf, hd.det_in_ch, hd.data_bytes, h->det));
    FRPCHK(parse_net(h, b, bytes, hd.emb_ops_offset, hd.n_emb_ops, hd.n_emb_bufs, hd.emb_in_buf, hd.emb_in_ch, hd.data_bytes, h->emb));
    for (int l = 0; l < 3; ++l)
        if (hd.det_head_buf[l] >= hd.n_det_bufs) return fail(h, FRP_ERR_BLOB, "head buffer id out of range");
    if (hd.emb_out_buf >= hd.n_emb_bufs) return fail(h, FRP_ERR_BLOB, "embedding buffer id out of range");
    // fp8-stored weights are expanded to fp16 behind the blob's data section (the kernels are the fp16 ones)
    std::vector<unsigned char> expanded;
    const unsigned char* data = b + hd.data_offset;
    size_t data_bytes = hd.data_bytes;
    bool any_fp8 = false;
    for (Net* net : {&h->det, &h->emb})
        for (const frp_conv_op& op : net->ops) any_fp8 |= (op.flags & FRP_OPFLAG_W_FP8) && !(op.flags & FRP_OPFLAG_FP8_MFMA);
    if (any_fp8) {
        expanded.assign(data, data + hd.data_bytes);
        for (Net* net : {&h->det, &h->emb})
            for (frp_conv_op& op : net->ops) {
                if (!(op.flags & FRP_OPFLAG_W_FP8) || (op.flags & FRP_OPFLAG_FP8_MFMA)) continue;   // fp8 ops use the bytes as stored
                const size_t per_row = (size_t)op.ksize * op.ksize * op.cin, n = per_row * op.cout;
                const size_t src = (size_t)op.w_off, sc = src + (n + 15) / 16 * 16;
                size_t dst = (expanded.size() + 255) / 256 * 256;
                expanded.resize(dst + n * 2);
                uint16_t* out16 = reinterpret_cast<uint16_t*>(expanded.data() + dst);
                for (int r = 0; r < op.cout; ++r) {
                    float scale;
                    memcpy(&scale, data + sc + (size_t)r * 4, 4);
                    for (size_t i = 0; i < per_row; ++i)
                        out16[(size_t)r * per_row + i] = f32_to_f16_bits(fp8_e4m3_value(data[src + (size_t)r * per_row + i]) * scale);
                }
                op.w_off = (int64_t)dst;
                op.flags &= ~FRP_OPFLAG_W_FP8;
            }
        data = expanded.data();
        data_bytes = expanded.size();
    }
    // Winograd weight images (conv3x3_wino.hip) for the embedder's eligible 3x3 stride-1 layers, appended behind the data
    // section.  The embedder's geometry is static (112 x 112 chips), so the map width of every op is known here; the
    // detector's maps at camera resolutions are wider than the kernel's LDS holds.  FRP_NO_WINO=1: direct kernels only.
    h->det.wino_off.assign(h->det.ops.size(), -1);
    h->emb.wino_off.assign(h->emb.ops.size(), -1);
    if (!getenv("FRP_NO_WINO")) {
        if (expanded.empty()) expanded.assign(data, data + hd.data_bytes);
        std::vector<int> bw(h->emb.n_bufs, 0);
        bw[h->emb.in_buf] = FRP_CHIP;
        for (size_t i = 0; i < h->emb.ops.size(); ++i) {
            const frp_conv_op& op = h->emb.ops[i];
            const int win = (op.flags & FRP_FLAG_FLATTEN) ? 1 : bw[op.in_buf];
            const int wout = (win + 2 * (op.ksize / 2) - op.ksize) / op.stride + 1;
            bw[op.out_buf] = wout;
            if (op.out2_buf >= 0) bw[op.out2_buf] = wout;
            if (!conv3x3_wino_shape_ok(win, op.cin, op.ksize, op.stride) || op.cout < 64 ||
                (op.flags & (FRP_FLAG_OUT_F32 | FRP_FLAG_FLATTEN | FRP_FLAG_RES_UP2 | FRP_OPFLAG_W_FP8 | FRP_OPFLAG_FP8_MFMA | FRP_OPFLAG_OUT_FP8)) ||
                op.out2_buf >= 0)
                continue;
            const size_t bytes = conv3x3_wino_image_bytes(op.cin, op.cout);
            const size_t dst = (expanded.size() + 255) / 256 * 256;
            expanded.resize(dst + bytes);
            build_wino_image(reinterpret_cast<const uint16_t*>(expanded.data() + op.w_off), op.cin, op.cout,
                             reinterpret_cast<uint16_t*>(expanded.data() + dst));
            h->emb.wino_off[i] = (int64_t)dst;
        }
        // The detector's maps depend on the frame size, so which of its layers take the kernel (in its 2-D tile form: maps wider than
        // 30 pixels) is decided per launch (conv3x3_wino.hip: wino_2d_pays); every 3x3 stride-1 layer of 128 channels and more that
        // could gets an image here (a third more weight bytes for those layers).
        for (size_t i = 0; i < h->det.ops.size(); ++i) {
            const frp_conv_op& op = h->det.ops[i];
            if (op.ksize != 3 || op.stride != 1 || (op.cin & 63) || op.cin < 128 || op.cout < 64 || op.out2_buf >= 0 ||
                (op.flags & (FRP_FLAG_OUT_F32 | FRP_FLAG_FLATTEN | FRP_FLAG_RES_UP2 | FRP_OPFLAG_W_FP8 | FRP_OPFLAG_FP8_MFMA | FRP_OPFLAG_OUT_FP8)))
                continue;
            const size_t bytes = conv3x3_wino_image_bytes(op.cin, op.cout);
            const size_t dst = (expanded.size() + 255) / 256 * 256;
            expanded.resize(dst + bytes);
            build_wino_image(reinterpret_cast<const uint16_t*>(expanded.data() + op.w_off), op.cin, op.cout,
                             reinterpret_cast<uint16_t*>(expanded.data() + dst));
            h->det.wino_off[i] = (int64_t)dst;
        }
        data = expanded.data();
        data_bytes = expanded.size();
    }
    // K-concat plans (both networks; FRP_NO_KCONCAT=1: every op as written in the blob)
    for (Net* net : {&h->det, &h->emb}) {
        const size_t n_ops = net->ops.size();
        net->kc_skip.assign(n_ops, 0);
        net->kc_src.assign(n_ops, -1);
        net->kc_w_off.assign(n_ops, -1);
        net->kc_bias_off.assign(n_ops, -1);
        if (getenv("FRP_NO_KCONCAT")) continue;
        for (size_t j = 0; j < n_ops; ++j) {
            const frp_conv_op& c = net->ops[j];
            // consumer: 3x3 conv over whole channel blocks with a plain residual, fp16 operands, one bias class
            if (c.ksize != 3 || c.res_buf < 0 || (c.cin & 63) || c.flags != 0) continue;
            // producer of the residual: the last writer of res_buf before j
            int i = -1;
            for (int q = (int)j - 1; q >= 0; --q)
                if (net->ops[q].out_buf == c.res_buf || net->ops[q].out2_buf == c.res_buf) { i = q; break; }
            if (i < 0) continue;
            const frp_conv_op& d = net->ops[i];
            if (d.out_buf != c.res_buf || d.ksize != 1 || d.stride != c.stride || d.act != FRP_ACT_NONE || d.res_buf >= 0 || d.flags != 0 ||
                d.out2_buf >= 0 || d.cout != c.cout || (d.cin & 63) || !(c.cin == d.cin || c.cin == 2 * d.cin) || net->kc_skip[i])
                continue;
            // the shortcut map has no other reader while it holds this tensor; its input and the consumer's input stay
            // untouched from the shortcut op to the consumer, and the consumer does not write over the shortcut's input
            bool ok = c.out_buf != d.in_buf && c.in_buf != d.in_buf;
            for (size_t q = (size_t)i + 1; q < n_ops && ok; ++q) {
                const frp_conv_op& o = net->ops[q];
                if (q != j && (o.in_buf == c.res_buf || o.res_buf == c.res_buf)) ok = false;     // another reader
                if (q < j && (o.out_buf == d.in_buf || o.out2_buf == d.in_buf)) ok = false;       // shortcut input rewritten early
                if (o.out_buf == c.res_buf || o.out2_buf == c.res_buf) break;                     // the buffer moves on to another tensor
            }
            if (!ok) continue;
            if (net == &h->det) { for (int l = 0; l < 3; ++l) ok &= (int)hd.det_head_buf[l] != c.res_buf; }   // (read by the decode kernel)
            else ok &= (int)hd.emb_out_buf != c.res_buf;
            if (!ok) continue;
            if (expanded.empty()) expanded.assign(data, data + hd.data_bytes);
            const size_t k1 = (size_t)9 * c.cin, k2 = (size_t)d.cin, kt = k1 + k2;
            const size_t wdst = (expanded.size() + 255) / 256 * 256;
            expanded.resize(wdst + (size_t)c.cout * kt * 2);
            const size_t bdst = (expanded.size() + 255) / 256 * 256;
            expanded.resize(bdst + (size_t)c.cout * 4);
            for (int r = 0; r < c.cout; ++r) {
                memcpy(expanded.data() + wdst + ((size_t)r * kt) * 2, expanded.data() + c.w_off + (size_t)r * k1 * 2, k1 * 2);
                memcpy(expanded.data() + wdst + ((size_t)r * kt + k1) * 2, expanded.data() + d.w_off + (size_t)r * k2 * 2, k2 * 2);
                float b1, b2;
                memcpy(&b1, expanded.data() + c.bias_off + (size_t)r * 4, 4);
                memcpy(&b2, expanded.data() + d.bias_off + (size_t)r * 4, 4);
                const float bs = b1 + b2;
                memcpy(expanded.data() + bdst + (size_t)r * 4, &bs, 4);
            }
            net->kc_skip[i] = 1;
            net->kc_src[j] = i;
            net->kc_w_off[j] = (int64_t)wdst;
            net->kc_bias_off[j] = (int64_t)bdst;
            data = expanded.data();
            data_bytes = expanded.size();
        }
    }
    FRPCHK(ensure(h, h->wdata, data_bytes));
    HIPCHK(h, hipMemcpyAsync(h->wdata.p, data, data_bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->hdr = hd;
    h->have_weights = true;
    return FRP_OK;
}

// ---------------------------------------------------------------- gallery
int frp_gallery_set(frp_handle* h, const void* emb, int64_t n, int32_t d, int32_t dtype) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (h->g_reserved.p) return fail(h, FRP_ERR_INVALID, "a gallery reservation is pending: commit or cancel it first (frp_gallery_commit / frp_gallery_cancel)");
    if (n < 0 || (n > 0 && !emb) || d != FRP_EMB_DIM) return fail(h, FRP_ERR_INVALID, "gallery must be [n x 512]");
    DevBuf fresh, fresh_x;   // new snapshot(s), swapped in when complete
    if (n > 0) {
        std::vector<float> f;
        const float* rows = (const float*)emb;
        if (dtype != FRP_F32) { FRPCHK(to_f32(h, emb, (size_t)n * d, dtype, f)); rows = f.data(); }
        FRPCHK(ensure(h, fresh, (size_t)n * d * 2));
        int r = upload_rows_normalized(h, rows, n, (_Float16*)fresh.p);
        if (r == FRP_OK && h->g_exact) {
            r = ensure(h, fresh_x, (size_t)n * d * 8);
            if (r == FRP_OK) r = upload_rows_exact(h, emb, n, dtype, (double*)fresh_x.p);
        }
        if (r != FRP_OK) { release(fresh); release(fresh_x); return r; }
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    release(h->gallery);
    h->gallery = fresh;
    release(h->gx);
    h->gx = fresh_x;
    h->g_rows = n;
    return FRP_OK;
}

int frp_gallery_set_device(frp_handle* h, const void* dev_f16, int64_t n, int32_t d) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (h->g_reserved.p) return fail(h, FRP_ERR_INVALID, "a gallery reservation is pending: commit or cancel it first (frp_gallery_commit / frp_gallery_cancel)");
    if (n <= 0 || !dev_f16 || d != FRP_EMB_DIM) return fail(h, FRP_ERR_INVALID, "gallery must be [n x 512] fp16 on the device");
    DevBuf fresh;
    FRPCHK(ensure(h, fresh, (size_t)n * d * 2));
    hipError_t e = hipMemcpyAsync(fresh.p, dev_f16, (size_t)n * d * 2, hipMemcpyDeviceToDevice, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { release(fresh); return fail(h, FRP_ERR_HIP, std::string("gallery copy: ") + hipGetErrorString(e)); }
    DevBuf fresh_x;
    if (h->g_exact) { int r = exact_from_f16(h, fresh.p, n, fresh_x); if (r != FRP_OK) { release(fresh); return r; } }
    release(h->gallery);
    h->gallery = fresh;
    release(h->gx);
    h->gx = fresh_x;
    h->g_rows = n;
    return FRP_OK;
}

int frp_gallery_reserve(frp_handle* h, int64_t capacity_rows, void** dev_f16) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (!dev_f16 || capacity_rows <= 0 || capacity_rows > 0x7fffff00L) return fail(h, FRP_ERR_INVALID, "bad gallery reservation");
    release(h->g_reserved);
    FRPCHK(ensure(h, h->g_reserved, (size_t)capacity_rows * FRP_EMB_DIM * 2));
    *dev_f16 = h->g_reserved.p;
    return FRP_OK;
}

int frp_gallery_cancel(frp_handle* h) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (h->g_reserved.p) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        release(h->g_reserved);
    }
    return FRP_OK;
}

int frp_gallery_commit(frp_handle* h, int64_t n_rows) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (!h->g_reserved.p || n_rows < 0 || (size_t)n_rows * FRP_EMB_DIM * 2 > h->g_reserved.cap)
        return fail(h, FRP_ERR_INVALID, "gallery commit without a matching reservation");
    HIPCHK(h, hipStreamSynchronize(h->stream));      // nothing of this handle still reads the old snapshot
    DevBuf fresh_x;
    if (h->g_exact) FRPCHK(exact_from_f16(h, h->g_reserved.p, n_rows, fresh_x));
    release(h->gallery);
    h->gallery = h->g_reserved;
    h->g_reserved = DevBuf();
    release(h->gx);
    h->gx = fresh_x;
    h->g_rows = n_rows;
    return FRP_OK;
}

// ---------------------------------------------------------------- multi-GPU: the one collective of the path, on RCCL
// frames are sharded one stream per GPU and need no exchange; the watch list is the exception - every rank decrypts /
// builds N / R rows and the full unit fp16 matrix is all-gathered over xGMI at load and on updates (the reference holds ENCODINGS
// once, in its one process: backend/app/state.py:78).  The library owns that collective: librccl is opened at first use (dlopen - a
// process that never goes multi-GPU does not load it), the communicator lives in the handle, and the gather lands STRAIGHT in a
// reserved snapshot (shard r at row offset r * ceil(N / R): no compaction copy) that is then committed like any other gallery.
// The caller's launcher (torch.distributed.run, MPI, a shell loop) only has to carry the 128-byte unique id from rank 0 to the others.
static_assert(sizeof(ncclUniqueId) == FRP_DIST_ID_BYTES, "include/frp.h: FRP_DIST_ID_BYTES");

int frp_dist_unique_id(void* id128) {
    if (!id128) return FRP_ERR_INVALID;
    Rccl& r = rccl();
    if (!r.err.empty()) return FRP_ERR_HIP;
    ncclUniqueId id;
    if (r.GetUniqueId(&id) != ncclSuccess) return FRP_ERR_HIP;
    memcpy(id128, &id, sizeof(id));
    return FRP_OK;
}

int frp_dist_init(frp_handle* h, const void* id128, int32_t rank, int32_t world) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (!id128 || world <= 0 || rank < 0 || rank >= world) return fail(h, FRP_ERR_INVALID, "bad rank / world size");
    if (h->comm) return fail(h, FRP_ERR_INVALID, "this handle already has a communicator (frp_dist_destroy first)");
    Rccl& r = rccl();
    if (!r.err.empty()) return fail(h, FRP_ERR_HIP, r.err);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    const ncclResult_t e = r.CommInitRank(&c, world, id, rank);          // collective over the ranks (the guard has set this handle's device)
    if (e != ncclSuccess) return fail(h, FRP_ERR_HIP, std::string("ncclCommInitRank: ") + r.GetErrorString(e));
    h->comm = c;
    h->dist_rank = rank;
    h->dist_world = world;
    return FRP_OK;
}

int frp_dist_destroy(frp_handle* h) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (h->comm) {
        (void)hipStreamSynchronize(h->stream);
        (void)rccl().CommDestroy((ncclComm_t)h->comm);
        h->comm = nullptr;
        h->dist_world = 0;
    }
    return FRP_OK;
}

int frp_gallery_allgather(frp_handle* h, const void* shard, int64_t shard_rows, int32_t dtype, int64_t n_total) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (!h->comm) return fail(h, FRP_ERR_INVALID, "no communicator (frp_dist_init)");
    if (h->g_reserved.p) return fail(h, FRP_ERR_INVALID, "a gallery reservation is pending: commit or cancel it first");
    const int world = h->dist_world, rank = h->dist_rank;
    if (n_total <= 0 || n_total > 0x7fffff00L || shard_rows < 0 || (shard_rows > 0 && !shard)) return fail(h, FRP_ERR_INVALID, "bad shard");
    const int64_t block = (n_total + world - 1) / world;
    const int64_t first = std::min<int64_t>((int64_t)rank * block, n_total), mine = std::min<int64_t>(block, n_total - first);
    if (shard_rows != mine) return fail(h, FRP_ERR_INVALID, "this rank owns rows [rank * ceil(N / R), ...): shard has another row count");
    Rccl& r = rccl();
    // this rank's rows, unit fp16, padded with zero rows to the block size (the last ranks' shards may be short or empty)
    DevBuf send;
    FRPCHK(ensure(h, send, (size_t)block * FRP_EMB_DIM * 2));
    int rc = FRP_OK;
    hipError_t he = hipMemsetAsync(send.p, 0, (size_t)block * FRP_EMB_DIM * 2, h->stream);
    if (he != hipSuccess) rc = fail(h, FRP_ERR_HIP, std::string("memset: ") + hipGetErrorString(he));
    if (rc == FRP_OK && mine > 0) {
        std::vector<float> f;
        const float* rows = (const float*)shard;
        if (dtype != FRP_F32) { rc = to_f32(h, shard, (size_t)mine * FRP_EMB_DIM, dtype, f); rows = f.data(); }
        if (rc == FRP_OK) rc = upload_rows_normalized(h, rows, mine, (_Float16*)send.p);
    }
    if (rc == FRP_OK) rc = ensure(h, h->g_reserved, (size_t)world * block * FRP_EMB_DIM * 2);
    if (rc == FRP_OK) {
        const ncclResult_t e = r.AllGather(send.p, h->g_reserved.p, (size_t)block * FRP_EMB_DIM, ncclFloat16, (ncclComm_t)h->comm, h->stream);
        if (e != ncclSuccess) rc = fail(h, FRP_ERR_HIP, std::string("ncclAllGather: ") + r.GetErrorString(e));
    }
    if (rc == FRP_OK) {
        he = hipStreamSynchronize(h->stream);
        if (he != hipSuccess) rc = fail(h, FRP_ERR_HIP, std::string("all-gather: ") + hipGetErrorString(he));
    }
    release(send);
    if (rc != FRP_OK) { release(h->g_reserved); return rc; }
    // commit (as frp_gallery_commit): rows [0, n_total) of the gathered blocks ARE the gallery
    DevBuf fresh_x;
    if (h->g_exact) {
        rc = exact_from_f16(h, h->g_reserved.p, n_total, fresh_x);
        if (rc != FRP_OK) { release(h->g_reserved); return rc; }
    }
    release(h->gallery);
    h->gallery = h->g_reserved;
    h->g_reserved = DevBuf();
    release(h->gx);
    h->gx = fresh_x;
    h->g_rows = n_total;
    return FRP_OK;
}

const void* frp_gallery_device_ptr(frp_handle* h) {
    if (!h) return nullptr;
    Guard g(h);
    return h->g_rows > 0 ? h->gallery.p : nullptr;
}

int frp_gallery_update_row(frp_handle* h, int64_t row, const void* emb, int32_t d, int32_t dtype) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (h->g_reserved.p) return fail(h, FRP_ERR_INVALID, "a gallery reservation is pending: commit or cancel it first (frp_gallery_commit / frp_gallery_cancel)");
    if (!emb || d != FRP_EMB_DIM || row < 0 || row > h->g_rows) return fail(h, FRP_ERR_INVALID, "bad gallery row");
    std::vector<float> f;
    FRPCHK(to_f32(h, emb, (size_t)d, dtype, f));
    if (row == h->g_rows && (size_t)(h->g_rows + 1) * d * 2 > h->gallery.cap) {
        // grow: new snapshot with doubled capacity
        DevBuf fresh;
        const size_t cap_rows = std::max<int64_t>(1024, h->g_rows * 2);
        FRPCHK(ensure(h, fresh, cap_rows * d * 2));
        if (h->g_rows > 0) {
            hipError_t e = hipMemcpyAsync(fresh.p, h->gallery.p, (size_t)h->g_rows * d * 2, hipMemcpyDeviceToDevice, h->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) { release(fresh); return fail(h, FRP_ERR_HIP, std::string("gallery grow: ") + hipGetErrorString(e)); }
        }
        release(h->gallery);
        h->gallery = fresh;
    }
    if (h->g_exact && (size_t)(row + 1) * d * 8 > h->gx.cap) {        // the exact copy grows with the snapshot's row capacity
        DevBuf fresh;
        const size_t cap_rows = std::max<size_t>(h->gallery.cap / ((size_t)d * 2), (size_t)row + 1);
        FRPCHK(ensure(h, fresh, cap_rows * d * 8));
        if (h->g_rows > 0) {
            hipError_t e = hipMemcpyAsync(fresh.p, h->gx.p, (size_t)h->g_rows * d * 8, hipMemcpyDeviceToDevice, h->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) { release(fresh); return fail(h, FRP_ERR_HIP, std::string("exact gallery grow: ") + hipGetErrorString(e)); }
        }
        release(h->gx);
        h->gx = fresh;
    }
    FRPCHK(upload_rows_normalized(h, f.data(), 1, (_Float16*)h->gallery.p + row * d));
    if (h->g_exact) FRPCHK(upload_rows_exact(h, emb, 1, dtype, (double*)h->gx.p + row * d));
    if (row == h->g_rows) h->g_rows += 1;
    return FRP_OK;
}

int frp_gallery_remove_row(frp_handle* h, int64_t row) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (h->g_reserved.p) return fail(h, FRP_ERR_INVALID, "a gallery reservation is pending: commit or cancel it first (frp_gallery_commit / frp_gallery_cancel)");
    if (row < 0 || row >= h->g_rows) return fail(h, FRP_ERR_INVALID, "bad gallery row");
    const int64_t last = h->g_rows - 1;
    if (row != last) {
        HIPCHK(h, hipMemcpyAsync((_Float16*)h->gallery.p + row * FRP_EMB_DIM, (_Float16*)h->gallery.p + last * FRP_EMB_DIM,
                                 FRP_EMB_DIM * 2, hipMemcpyDeviceToDevice, h->stream));
        if (h->g_exact)
            HIPCHK(h, hipMemcpyAsync((double*)h->gx.p + row * FRP_EMB_DIM, (double*)h->gx.p + last * FRP_EMB_DIM, FRP_EMB_DIM * 8,
                                     hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    h->g_rows = last;
    return FRP_OK;
}

int64_t frp_gallery_size(const frp_handle* h) { return h ? h->g_rows : -1; }

int frp_gallery_get(frp_handle* h, void* out_f16, int64_t first_row, int64_t n_rows) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (!out_f16 || first_row < 0 || n_rows < 0 || first_row + n_rows > h->g_rows) return fail(h, FRP_ERR_INVALID, "bad gallery range");
    if (n_rows == 0) return FRP_OK;
    HIPCHK(h, hipMemcpyAsync(out_f16, (_Float16*)h->gallery.p + first_row * FRP_EMB_DIM, (size_t)n_rows * FRP_EMB_DIM * 2,
                             hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return FRP_OK;
}

int frp_gallery_exact(frp_handle* h, int32_t on) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (!on) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        release(h->gx); release(h->gx_q); release(h->gx_out);
        h->g_exact = false;
        return FRP_OK;
    }
    if (h->g_exact) return FRP_OK;
    DevBuf fresh;
    if (h->g_rows > 0) {          // rows that exist already: the unit fp16 rows widened (their exact values are gone)
        const size_t cap_rows = std::max<size_t>(h->gallery.cap / ((size_t)FRP_EMB_DIM * 2), (size_t)h->g_rows);
        FRPCHK(ensure(h, fresh, cap_rows * FRP_EMB_DIM * 8));
        hipError_t e = launch_gallery_widen((const _Float16*)h->gallery.p, (double*)fresh.p, h->g_rows, FRP_EMB_DIM, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { release(fresh); return fail(h, FRP_ERR_HIP, std::string("gallery_widen: ") + hipGetErrorString(e)); }
    }
    release(h->gx);
    h->gx = fresh;
    h->g_exact = true;
    return FRP_OK;
}

int frp_gallery_distances(frp_handle* h, const double* q, int32_t M, double* dist, int64_t n_cols) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (!h->g_exact) return fail(h, FRP_ERR_INVALID, "exact rows are not enabled (frp_gallery_exact)");
    if (!q || !dist || M <= 0 || M > 65536) return fail(h, FRP_ERR_INVALID, "bad distance arguments");
    if (n_cols != h->g_rows) return fail(h, FRP_ERR_INVALID, "gallery_distances: output sized for another gallery size");
    if (h->g_rows == 0) return FRP_OK;
    FRPCHK(ensure(h, h->gx_q, (size_t)M * FRP_EMB_DIM * 8));
    FRPCHK(ensure(h, h->gx_out, (size_t)M * h->g_rows * 8));
    HIPCHK(h, hipMemcpyAsync(h->gx_q.p, q, (size_t)M * FRP_EMB_DIM * 8, hipMemcpyHostToDevice, h->stream));
    hipError_t e = launch_gallery_distances((const double*)h->gx.p, h->g_rows, (const double*)h->gx_q.p, M, (double*)h->gx_out.p, h->stream);
    if (e != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("gallery_distances: ") + hipGetErrorString(e));
    HIPCHK(h, hipMemcpyAsync(dist, h->gx_out.p, (size_t)M * h->g_rows * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return FRP_OK;
}

int frp_gallery_get_exact(frp_handle* h, double* out, int64_t first_row, int64_t n_rows) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (!h->g_exact) return fail(h, FRP_ERR_INVALID, "exact rows are not enabled (frp_gallery_exact)");
    if (!out || first_row < 0 || n_rows < 0 || first_row + n_rows > h->g_rows) return fail(h, FRP_ERR_INVALID, "bad gallery range");
    if (n_rows == 0) return FRP_OK;
    HIPCHK(h, hipMemcpyAsync(out, (double*)h->gx.p + first_row * FRP_EMB_DIM, (size_t)n_rows * FRP_EMB_DIM * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return FRP_OK;
}

// ---------------------------------------------------------------- hot path
int frp_upload_frames(frp_handle* h, const uint8_t* bgr, int32_t B, int32_t H, int32_t W, int64_t row_stride) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    FRPCHK(upload_frames(h, bgr, B, H, W, row_stride));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return FRP_OK;
}

void* frp_host_alloc(frp_handle* h, size_t bytes) {
    if (!h || bytes == 0) return nullptr;
    Guard g(h);
    void* p = nullptr;
    if (hipSetDevice(h->device) != hipSuccess || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        fail(h, FRP_ERR_OOM, "hipHostMalloc failed");
        return nullptr;
    }
    h->pinned.push_back(p);
    return p;
}

void frp_host_free(frp_handle* h, void* p) {
    if (!h || !p) return;
    Guard g(h);
    for (size_t i = 0; i < h->pinned.size(); ++i)
        if (h->pinned[i] == p) {
            (void)hipStreamSynchronize(h->copy_stream);
            (void)hipHostFree(p);
            h->pinned.erase(h->pinned.begin() + (long)i);
            return;
        }
}

int frp_upload_frames_async(frp_handle* h, const uint8_t* bgr, int32_t B, int32_t H, int32_t W, int64_t row_stride) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h, false);      // copy stream only: does not re-record the stage events, so a pending pass is not drained
                            // (with the timers on, settling here made the upload of batch t+1 wait for batch t)
    if (!bgr || B <= 0 || H <= 0 || W <= 0 || row_stride < (int64_t)W * 3) return fail(h, FRP_ERR_INVALID, "bad frame arguments");
    if (B > 1024) return fail(h, FRP_ERR_INVALID, "batch too large (max 1024 frames per call)");
    const size_t need = (size_t)B * H * W * 3;
    if (need > h->frames_next.cap || !h->frames_next.p) {
        // growing the staging buffer: nothing may still be copying into / computing from it
        HIPCHK(h, hipStreamSynchronize(h->copy_stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        FRPCHK(ensure(h, h->frames_next, need));
    }
    // the staging buffer was the resident one until the last swap: wait for the work enqueued before it
    HIPCHK(h, hipStreamWaitEvent(h->copy_stream, h->ev_next_free, 0));
    HIPCHK(h, hipMemcpy2DAsync(h->frames_next.p, (size_t)W * 3, bgr, (size_t)row_stride, (size_t)W * 3, (size_t)B * H,
                               hipMemcpyHostToDevice, h->copy_stream));
    HIPCHK(h, hipEventRecord(h->ev_next_ready, h->copy_stream));
    h->nB = B; h->nH = H; h->nW = W;
    h->next_valid = true;
    return FRP_OK;
}

int frp_jpeg_info_get(const uint8_t* data, size_t size, frp_jpeg_info* info) {
    if (!data || !info) return FRP_ERR_INVALID;
    return jpeg_info(data, size, info, nullptr);
}

int frp_jpeg_coefficients(const uint8_t* data, size_t size, int16_t* coef, size_t coef_elems, uint16_t* qtab, frp_jpeg_info* info) {
    if (!data || !coef || !qtab) return FRP_ERR_INVALID;
    return jpeg_decode_coefficients(data, size, coef, coef_elems, qtab, info, nullptr);
}

// Device entropy decode of a batch whose frames all carry restart intervals (round 5; jpeg_kernels.hip: jpeg_huffman_kernel): the host
// parses headers and finds the RSTn markers (one memchr pass), the COMPRESSED scans go to the device (~0.5 MB per 1080p frame instead of
// 6.3 MB of coefficients), one thread per interval decodes, and the host waits only for the per-image error flags (a corrupt stream
// must be reported by this call, as on the host path) before the pixel kernels are queued.  -> FRP_OK, an error, or 1 = "not this
// batch" (not switched on, no restart intervals, too few of them to fill a wave): the caller takes the host decoder.
namespace {
int upload_jpeg_device(frp_handle* h, const uint8_t* const* jpegs, const size_t* sizes, int32_t B, const frp_jpeg_info& I, int turn,
                       JpegParams& p, size_t total_host_layout, size_t q_off) {
    // When: one thread per interval decodes 32 x 1080p frames in 17.8 ms at one interval per MCU row (120 MCUs), 4.5 ms at 30 MCUs,
    // 1.25 ms at 8 (profiles/r5/jpeg_device_entropy.txt) - the time goes with the LENGTH of an interval, and 16 host threads take
    // 9-12 ms: by default the device decodes streams whose intervals are at most 32 MCUs and the host the others.
    // FRP_JPEG_DEVICE_HUFFMAN=1 (read once): the device whatever the interval (takes the entropy decode off the host's cores; at
    // one interval per row it is slower than the pipeline consumes frames), =0: never.
    static const int mode = [] { const char* e = getenv("FRP_JPEG_DEVICE_HUFFMAN"); return !e ? 0 : (e[0] == '0' ? -1 : 1); }();
    if (mode < 0 || I.restart_interval <= 0 || (mode == 0 && I.restart_interval > 32)) return 1;
    const long mcus = (long)I.mcus_x * I.mcus_y;
    const long n_int = (mcus + I.restart_interval - 1) / I.restart_interval;
    if ((long)B * n_int < 64 || n_int > 0x7fffff) return 1;
    std::vector<JpegDevicePlan> plans((size_t)B);
    std::vector<JpegHuffTableDev> tabs((size_t)B * 6);
    for (int i = 0; i < B; ++i) {
        std::string e;
        if (!jpegs[i]) return fail(h, FRP_ERR_INVALID, "JPEG " + std::to_string(i) + ": null image");
        const int rc = jpeg_plan_device_decode(jpegs[i], sizes[i], plans[i], tabs.data() + (size_t)i * 6, &e);
        if (rc != FRP_OK) {
            if (plans[i].info.restart_interval <= 0 && plans[i].info.width > 0) return 1;          // a frame without intervals: host path for the batch
            return fail(h, rc, "JPEG " + std::to_string(i) + ": " + e);
        }
        const frp_jpeg_info& Ii = plans[i].info;
        if (Ii.width != I.width || Ii.height != I.height || Ii.components != I.components || Ii.h_samp[0] != I.h_samp[0] || Ii.v_samp[0] != I.v_samp[0])
            return fail(h, FRP_ERR_INVALID, "JPEG " + std::to_string(i) + ": geometry differs from image 0 (one batch = one frame size and sampling)");
        if (Ii.restart_interval != I.restart_interval) return 1;
    }
    // staging layout: scans | interval offsets | tables | quantisation tables | error flags (read back)
    std::vector<size_t> soff((size_t)B + 1, 0);
    for (int i = 0; i < B; ++i) soff[i + 1] = (soff[i] + plans[i].scan_bytes + 15) & ~(size_t)15;
    if (soff[B] >= 0xfffffff0u) return 1;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_int = up(soff[B]), o_tab = up(o_int + (size_t)B * (n_int + 1) * 4), o_q = up(o_tab + (size_t)B * 6 * sizeof(JpegHuffTableDev)),
                 o_err = up(o_q + (size_t)B * 192 * 2), stage_total = o_err + (size_t)B * 4;
    if (h->jpeg_h2d_pending[turn]) {
        HIPCHK(h, hipEventSynchronize(h->ev_jpeg_h2d[turn]));
        h->jpeg_h2d_pending[turn] = false;
    }
    if (stage_total > h->jpeg_pin_cap[turn]) {
        if (h->jpeg_pin[turn]) { (void)hipHostFree(h->jpeg_pin[turn]); h->jpeg_pin[turn] = nullptr; h->jpeg_pin_cap[turn] = 0; }
        if (hipHostMalloc(&h->jpeg_pin[turn], stage_total, hipHostMallocDefault) != hipSuccess) return fail(h, FRP_ERR_OOM, "hipHostMalloc (JPEG staging) failed");
        h->jpeg_pin_cap[turn] = stage_total;
    }
    if (!h->ev_jpeg_h2d[turn]) HIPCHK(h, hipEventCreateWithFlags(&h->ev_jpeg_h2d[turn], hipEventDisableTiming));
    char* st = (char*)h->jpeg_pin[turn];
    uint32_t* io = (uint32_t*)(st + o_int);
    for (int i = 0; i < B; ++i) {
        memcpy(st + soff[i], plans[i].scan, plans[i].scan_bytes);
        for (long k = 0; k <= n_int; ++k) io[(size_t)i * (n_int + 1) + k] = (uint32_t)(soff[i] + plans[i].int_off[(size_t)k]);
        memcpy(st + o_q + (size_t)i * 384, plans[i].qtab, 384);
    }
    memcpy(st + o_tab, tabs.data(), tabs.size() * sizeof(JpegHuffTableDev));
    const size_t need = (size_t)B * I.height * I.width * 3;
    if (need > h->frames_next.cap || !h->frames_next.p || total_host_layout > h->jpeg_coef.cap || (size_t)B * p.plane_img > h->jpeg_planes.cap ||
        o_err > h->jpeg_scan.cap || (size_t)B * 4 > h->jpeg_err.cap) {
        HIPCHK(h, hipStreamSynchronize(h->copy_stream));     // growing buffers: nothing may still be copying into / computing from them
        HIPCHK(h, hipStreamSynchronize(h->stream));
        FRPCHK(ensure(h, h->frames_next, need));
        FRPCHK(ensure(h, h->jpeg_coef, total_host_layout));
        FRPCHK(ensure(h, h->jpeg_planes, (size_t)B * p.plane_img));
        FRPCHK(ensure(h, h->jpeg_scan, o_err));
        FRPCHK(ensure(h, h->jpeg_err, (size_t)B * 4));
    }
    HIPCHK(h, hipStreamWaitEvent(h->copy_stream, h->ev_next_free, 0));
    HIPCHK(h, hipMemcpyAsync(h->jpeg_scan.p, st, o_err, hipMemcpyHostToDevice, h->copy_stream));
    HIPCHK(h, hipMemsetAsync(h->jpeg_coef.p, 0, q_off, h->copy_stream));
    HIPCHK(h, hipMemcpyAsync((char*)h->jpeg_coef.p + q_off, (char*)h->jpeg_scan.p + o_q, (size_t)B * 384, hipMemcpyDeviceToDevice, h->copy_stream));
    HIPCHK(h, hipMemsetAsync(h->jpeg_err.p, 0, (size_t)B * 4, h->copy_stream));
    JpegHuffParams hp{};
    hp.scan = (const uint8_t*)h->jpeg_scan.p;
    hp.int_off = (const uint32_t*)((const char*)h->jpeg_scan.p + o_int);
    hp.tables = (const JpegHuffTableDev*)((const char*)h->jpeg_scan.p + o_tab);
    hp.coef = (int16_t*)h->jpeg_coef.p;
    hp.err = (int32_t*)h->jpeg_err.p;
    hp.coef_per_image = (long)jpeg_coef_elems(I);
    hp.B = B; hp.n_int = (int)n_int; hp.ri = I.restart_interval;
    hp.mcus_x = I.mcus_x; hp.mcus_y = I.mcus_y; hp.components = I.components;
    for (int c = 0; c < 3; ++c) { hp.hs[c] = I.h_samp[c]; hp.vs[c] = I.v_samp[c]; hp.bx[c] = p.bx[c]; hp.comp_off[c] = p.plane_off[c]; }
    hipError_t e = launch_jpeg_huffman(hp, h->copy_stream);
    if (e != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("jpeg huffman: ") + hipGetErrorString(e));
    HIPCHK(h, hipMemcpyAsync(st + o_err, h->jpeg_err.p, (size_t)B * 4, hipMemcpyDeviceToHost, h->copy_stream));
    HIPCHK(h, hipStreamSynchronize(h->copy_stream));          // the flags decide this call's return value (and the staging buffer is free again)
    const int32_t* flags = (const int32_t*)(st + o_err);
    for (int i = 0; i < B; ++i)
        if (flags[i]) return fail(h, FRP_ERR_INVALID, "JPEG " + std::to_string(i) + ": corrupt or truncated entropy-coded data");
    p.coef = (const int16_t*)h->jpeg_coef.p;
    p.qtab = (const uint16_t*)((const char*)h->jpeg_coef.p + q_off);
    p.planes = (uint8_t*)h->jpeg_planes.p;
    p.frames = (uint8_t*)h->frames_next.p;
    e = launch_jpeg_decode(p, h->copy_stream);
    if (e != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("jpeg decode: ") + hipGetErrorString(e));
    HIPCHK(h, hipEventRecord(h->ev_next_ready, h->copy_stream));
    h->nB = B; h->nH = I.height; h->nW = I.width;
    h->next_valid = true;
    h->ctr_jpeg_device_batches += 1;
    return FRP_OK;
}
}  // namespace

int frp_upload_jpeg_async(frp_handle* h, const uint8_t* const* jpegs, const size_t* sizes, int32_t B) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h, false);      // copy stream only (as frp_upload_frames_async)
    if (!jpegs || !sizes || B <= 0 || B > 1024) return fail(h, FRP_ERR_INVALID, "bad JPEG batch arguments");
    frp_jpeg_info I{};
    std::string err;
    if (!jpegs[0] || jpeg_info(jpegs[0], sizes[0], &I, &err) != FRP_OK) return fail(h, FRP_ERR_INVALID, "JPEG 0: " + err);
    const size_t ce = jpeg_coef_elems(I);
    // staging: [B] coefficients (int16) then [B][3][64] tables (uint16), 16-byte aligned parts
    const size_t coef_bytes = (size_t)B * ce * 2, q_off = (coef_bytes + 255) & ~(size_t)255, total = q_off + (size_t)B * 3 * 64 * 2;
    const int turn = h->jpeg_turn;
    h->jpeg_turn ^= 1;
    JpegParams p{};
    p.B = B; p.W = I.width; p.H = I.height; p.components = I.components;
    p.hs = I.h_samp[0]; p.vs = I.v_samp[0];
    p.cw = (I.width + p.hs - 1) / p.hs;
    p.ch = (I.height + p.vs - 1) / p.vs;
    long off = 0;
    for (int c = 0; c < I.components; ++c) {
        p.bx[c] = I.mcus_x * I.h_samp[c];
        p.by[c] = I.mcus_y * I.v_samp[c];
        p.blocks_per_image += p.bx[c] * p.by[c];
        p.plane_off[c] = off;
        off += (long)p.bx[c] * p.by[c] * 64;
    }
    p.plane_img = off;
    {   // restart-interval streams: entropy decode on the device
        const int dr = upload_jpeg_device(h, jpegs, sizes, B, I, turn, p, total, q_off);
        if (dr != 1) return dr;
    }
    if (h->jpeg_h2d_pending[turn]) {                 // the copy of the batch before the previous one read this staging buffer
        HIPCHK(h, hipEventSynchronize(h->ev_jpeg_h2d[turn]));
        h->jpeg_h2d_pending[turn] = false;
    }
    if (total > h->jpeg_pin_cap[turn]) {
        if (h->jpeg_pin[turn]) { (void)hipHostFree(h->jpeg_pin[turn]); h->jpeg_pin[turn] = nullptr; h->jpeg_pin_cap[turn] = 0; }
        if (hipHostMalloc(&h->jpeg_pin[turn], total, hipHostMallocDefault) != hipSuccess) return fail(h, FRP_ERR_OOM, "hipHostMalloc (JPEG staging) failed");
        h->jpeg_pin_cap[turn] = total;
    }
    if (!h->ev_jpeg_h2d[turn]) HIPCHK(h, hipEventCreateWithFlags(&h->ev_jpeg_h2d[turn], hipEventDisableTiming));
    int16_t* coef = (int16_t*)h->jpeg_pin[turn];
    uint16_t* qtab = (uint16_t*)((char*)h->jpeg_pin[turn] + q_off);
    // entropy decoding: one image per task on host threads (the images are independent; within one the bit stream is serial)
    std::vector<int> rcs((size_t)B, FRP_OK);
    std::vector<std::string> errs((size_t)B);
    {
        const int nth = std::max(1, std::min<int>(B, (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()))));
        std::atomic<int> next{0};
        auto work = [&]() {
            for (int i = next.fetch_add(1); i < B; i = next.fetch_add(1)) {
                frp_jpeg_info Ii{};
                if (!jpegs[i]) { rcs[i] = FRP_ERR_INVALID; errs[i] = "null image"; continue; }
                rcs[i] = jpeg_decode_coefficients(jpegs[i], sizes[i], coef + (size_t)i * ce, ce, qtab + (size_t)i * 192, &Ii, &errs[i]);
                if (rcs[i] == FRP_OK && (Ii.width != I.width || Ii.height != I.height || Ii.components != I.components ||
                                         Ii.h_samp[0] != I.h_samp[0] || Ii.v_samp[0] != I.v_samp[0])) {
                    rcs[i] = FRP_ERR_INVALID;
                    errs[i] = "geometry differs from image 0 (one batch = one frame size and sampling)";
                }
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nth; ++t) th.emplace_back(work);
        work();
        for (auto& t : th) t.join();
    }
    for (int i = 0; i < B; ++i)
        if (rcs[i] != FRP_OK) return fail(h, rcs[i], "JPEG " + std::to_string(i) + ": " + errs[i]);
    const size_t need = (size_t)B * I.height * I.width * 3;
    if (need > h->frames_next.cap || !h->frames_next.p || total > h->jpeg_coef.cap || (size_t)B * off > h->jpeg_planes.cap) {
        HIPCHK(h, hipStreamSynchronize(h->copy_stream));     // growing buffers: nothing may still be copying into / computing from them
        HIPCHK(h, hipStreamSynchronize(h->stream));
        FRPCHK(ensure(h, h->frames_next, need));
        FRPCHK(ensure(h, h->jpeg_coef, total));
        FRPCHK(ensure(h, h->jpeg_planes, (size_t)B * off));
    }
    HIPCHK(h, hipStreamWaitEvent(h->copy_stream, h->ev_next_free, 0));     // the staging frame buffer was the resident one until the last swap
    HIPCHK(h, hipMemcpyAsync(h->jpeg_coef.p, h->jpeg_pin[turn], total, hipMemcpyHostToDevice, h->copy_stream));
    HIPCHK(h, hipEventRecord(h->ev_jpeg_h2d[turn], h->copy_stream));
    h->jpeg_h2d_pending[turn] = true;
    p.coef = (const int16_t*)h->jpeg_coef.p;
    p.qtab = (const uint16_t*)((const char*)h->jpeg_coef.p + q_off);
    p.planes = (uint8_t*)h->jpeg_planes.p;
    p.frames = (uint8_t*)h->frames_next.p;
    hipError_t e = launch_jpeg_decode(p, h->copy_stream);
    if (e != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("jpeg decode: ") + hipGetErrorString(e));
    HIPCHK(h, hipEventRecord(h->ev_next_ready, h->copy_stream));
    h->nB = B; h->nH = I.height; h->nW = I.width;
    h->next_valid = true;
    return FRP_OK;
}

// diagnostic: how many frp_upload_jpeg_async batches had their entropy decode on the device (restart-interval streams)
int64_t frp_debug_graph_replays(frp_handle* h) {
    if (!h) return -1;
    Guard g(h);
    return h->graph_replays;
}

int64_t frp_debug_jpeg_device_batches(frp_handle* h) {
    if (!h) return -1;
    Guard g(h, false);
    return h->ctr_jpeg_device_batches;
}

int frp_swap_frames(frp_handle* h) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h, false);      // enqueues a wait + an event on the compute stream; the stage events stay as recorded
    if (!h->next_valid) return fail(h, FRP_ERR_INVALID, "no staged frames (call frp_upload_frames_async)");
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_next_ready, 0));     // compute waits for the staged copy
    std::swap(h->frames, h->frames_next);
    HIPCHK(h, hipEventRecord(h->ev_next_free, h->stream));            // ... and the old resident buffer is free after
    h->rB = h->nB; h->rH = h->nH; h->rW = h->nW;                       // everything enqueued so far
    h->dH = h->rH; h->dW = h->rW; h->det_scaled = false;
    h->canvas_h = round_up(h->rH, 32);
    h->canvas_w = round_up(h->rW, 32);
    h->next_valid = false;
    return FRP_OK;
}

int frp_process_resident(frp_handle* h, int32_t max_faces, float det_thresh, float nms_iou, uint32_t flags) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    rec(h, EV_H2D);
    FRPCHK(run_pipeline(h, max_faces, det_thresh, nms_iou, flags));
    h->ev_pending = h->cfg.profile != 0;
    h->ctr.calls += 1;
    return FRP_OK;
}

int frp_synchronize(frp_handle* h) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h, false);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    FRPCHK(resolve_count(h));
    settle_events(h, true);
    return FRP_OK;
}

int frp_fetch_results(frp_handle* h, int32_t B, int32_t max_faces, float* boxes, float* kps, float* scores, int32_t* counts,
                      float* emb, int32_t* match_idx, float* match_cos) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h, false);
    // the caller sized its buffers for B x max_faces: refuse when another thread's call on this handle changed
    // the shape of the results in between (checked under the handle mutex)
    if (B != h->last_B || max_faces != h->last_K)
        return fail(h, FRP_ERR_INVALID, "fetch_results: buffers sized for another batch (results were replaced by a later call)");
    return fetch_results(h, boxes, kps, scores, counts, emb, match_idx, match_cos);
}

int frp_process_frames(frp_handle* h, const uint8_t* bgr, int32_t B, int32_t H, int32_t W, int64_t row_stride,
                       int32_t max_faces, float det_thresh, float nms_iou, uint32_t flags, float* boxes, float* kps,
                       float* scores, int32_t* counts, float* emb, int32_t* match_idx, float* match_cos) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    FRPCHK(upload_frames(h, bgr, B, H, W, row_stride));
    FRPCHK(run_pipeline(h, max_faces, det_thresh, nms_iou, flags));
    FRPCHK(fetch_results(h, boxes, kps, scores, counts, emb, match_idx, match_cos));
    accumulate_events(h, true);
    h->ctr.calls += 1;
    return FRP_OK;
}

int frp_detect(frp_handle* h, const uint8_t* bgr, int32_t B, int32_t H, int32_t W, int64_t row_stride, int32_t max_faces,
               float det_thresh, float nms_iou, uint32_t flags, float* boxes, float* kps, float* scores, int32_t* counts,
               int32_t* anchor_idx) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    FRPCHK(upload_frames(h, bgr, B, H, W, row_stride));
    FRPCHK(run_detect(h, max_faces, det_thresh, nms_iou, flags));
    const size_t s = (size_t)B * max_faces;
    if (boxes) HIPCHK(h, hipMemcpyAsync(boxes, h->boxes.p, s * 16, hipMemcpyDeviceToHost, h->stream));
    if (kps) HIPCHK(h, hipMemcpyAsync(kps, h->kps.p, s * 40, hipMemcpyDeviceToHost, h->stream));
    if (scores) HIPCHK(h, hipMemcpyAsync(scores, h->scores.p, s * 4, hipMemcpyDeviceToHost, h->stream));
    if (counts) HIPCHK(h, hipMemcpyAsync(counts, h->counts.p, (size_t)B * 4, hipMemcpyDeviceToHost, h->stream));
    if (anchor_idx) HIPCHK(h, hipMemcpyAsync(anchor_idx, h->anchor.p, s * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->last_nfaces = 0;
    return FRP_OK;
}

int frp_detect_resident(frp_handle* h, int32_t B, int32_t det_h, int32_t det_w, int32_t max_faces, float det_thresh, float nms_iou,
                        uint32_t flags, float* boxes, float* kps, float* scores, int32_t* counts, int32_t* anchor_idx) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (B != h->rB) return fail(h, FRP_ERR_INVALID, "detect_resident: buffers sized for another resident batch");
    rec(h, EV_H2D);
    FRPCHK(select_det_source(h, det_h, det_w));
    FRPCHK(run_detect(h, max_faces, det_thresh, nms_iou, flags));
    const size_t s = (size_t)B * max_faces;
    if (boxes) HIPCHK(h, hipMemcpyAsync(boxes, h->boxes.p, s * 16, hipMemcpyDeviceToHost, h->stream));
    if (kps) HIPCHK(h, hipMemcpyAsync(kps, h->kps.p, s * 40, hipMemcpyDeviceToHost, h->stream));
    if (scores) HIPCHK(h, hipMemcpyAsync(scores, h->scores.p, s * 4, hipMemcpyDeviceToHost, h->stream));
    if (counts) HIPCHK(h, hipMemcpyAsync(counts, h->counts.p, (size_t)B * 4, hipMemcpyDeviceToHost, h->stream));
    if (anchor_idx) HIPCHK(h, hipMemcpyAsync(anchor_idx, h->anchor.p, s * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    accumulate_detect_events(h);
    h->last_nfaces = 0;
    return FRP_OK;
}

int frp_get_det_source(frp_handle* h, uint8_t* out, int64_t out_bytes, int32_t* hs, int32_t* ws) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (h->rB <= 0) return fail(h, FRP_ERR_INVALID, "no resident frames");
    if (hs) *hs = h->dH;
    if (ws) *ws = h->dW;
    const int64_t need = (int64_t)h->rB * h->dH * h->dW * 3;
    if (!out) return FRP_OK;
    if (out_bytes < need) return fail(h, FRP_ERR_INVALID, "buffer too small");
    HIPCHK(h, hipMemcpyAsync(out, h->det_scaled ? h->scaled.p : h->frames.p, (size_t)need, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return FRP_OK;
}

int frp_finish_faces(frp_handle* h, int32_t B_in, const float* boxes, const float* kps, const float* scores, const int32_t* counts,
                     int32_t max_faces, uint32_t flags, float* emb, int32_t* match_idx, float* match_cos) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (!h->have_weights) return fail(h, FRP_ERR_NO_WEIGHTS, "no weights loaded");
    if (h->rB <= 0) return fail(h, FRP_ERR_INVALID, "no resident frames (call frp_upload_frames)");
    if (B_in != h->rB) return fail(h, FRP_ERR_INVALID, "finish_faces: face list sized for another resident batch");
    if (!kps || !counts || max_faces <= 0 || max_faces > FRP_MAX_FACES_CAP) return fail(h, FRP_ERR_INVALID, "bad face list");
    const int B = h->rB, K = max_faces;
    int n = 0;
    for (int b = 0; b < B; ++b) {
        if (counts[b] < 0 || counts[b] > K) return fail(h, FRP_ERR_INVALID, "face count out of range");
        n += counts[b];
    }
    FRPCHK(ensure_results(h, B, K));
    const size_t s = (size_t)B * K;
    HIPCHK(h, hipMemcpyAsync(h->kps.p, kps, s * 40, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->counts.p, counts, (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
    if (boxes) HIPCHK(h, hipMemcpyAsync(h->boxes.p, boxes, s * 16, hipMemcpyHostToDevice, h->stream));
    if (scores) HIPCHK(h, hipMemcpyAsync(h->scores.p, scores, s * 4, hipMemcpyHostToDevice, h->stream));
    hipError_t e = launch_compact_faces((const int32_t*)h->counts.p, B, K, (int32_t*)h->face_slot.p, (int32_t*)h->nfaces.p, h->stream);
    if (e != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("compact_faces: ") + hipGetErrorString(e));
    h->last_B = B;
    h->last_K = K;
    rec(h, EV_DEC);
    FRPCHK(run_faces(h, K, n, flags));
    FRPCHK(fetch_results(h, nullptr, nullptr, nullptr, nullptr, emb, match_idx, match_cos));
    accumulate_face_events(h);
    h->ctr.calls += 1;
    return FRP_OK;
}

int frp_get_head_map(frp_handle* h, int32_t level, void* out_f16, int64_t out_bytes, int32_t* hl, int32_t* wl) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (!h->have_weights || level < 0 || level > 2 || h->last_B <= 0) return fail(h, FRP_ERR_INVALID, "no head map available");
    const int bi = (int)h->hdr.det_head_buf[level];
    const TensorDims d = h->det.dims[bi];
    if (hl) *hl = d.h;
    if (wl) *wl = d.w;
    const int64_t need = (int64_t)h->last_B * d.h * d.w * d.c * 2;
    if (!out_f16) return FRP_OK;
    if (out_bytes < need) return fail(h, FRP_ERR_INVALID, "head map buffer too small");
    HIPCHK(h, hipMemcpyAsync(out_f16, h->det.bufs[bi].p, (size_t)need, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return FRP_OK;
}

// Diagnostic (tools/det_hash_bisect.py): enable != 0 - every later detector pass takes a 64-bit hash of each op's output right behind
// the op (stream order: the tensor as the NEXT op reads it); out64 (64 slots, op order) receives the hashes of the last pass.
int frp_debug_det_hashes(frp_handle* h, int32_t enable, uint64_t* out64) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (enable) {
        FRPCHK(ensure(h, h->det_hashes, 64 * 8));
        h->det_hash_on = true;
    } else h->det_hash_on = false;
    if (out64) {
        if (!h->det_hashes.p) return fail(h, FRP_ERR_INVALID, "hashes were never enabled");
        HIPCHK(h, hipMemcpyAsync(out64, h->det_hashes.p, 64 * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return FRP_OK;
}

// Diagnostic (tools/det_bisect.py): the detector program on the resident frames up to and including op `n_ops - 1`, then that op's
// output tensor [B, th, tw, tc] fp16.  Physical buffers are shared between tensors by liveness, so a tensor can only be read
// while nothing behind it has run: hence a prefix run rather than a read after a full pass.
int frp_debug_det_prefix(frp_handle* h, int32_t n_ops, void* out_f16, int64_t out_bytes, int32_t* th, int32_t* tw, int32_t* tc) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (!h->have_weights) return fail(h, FRP_ERR_NO_WEIGHTS, "no weights loaded");
    if (h->rB <= 0) return fail(h, FRP_ERR_INVALID, "no resident frames (call frp_upload_frames)");
    if (n_ops <= 0 || n_ops > (int)h->det.ops.size()) return fail(h, FRP_ERR_INVALID, "op count out of range");
    const int B = h->rB, Hc = h->canvas_h, Wc = h->canvas_w;
    const bool fused = stem_fusable(h->det) && !getenv("FRP_NO_FUSED_STEM");
    if (!fused) return fail(h, FRP_ERR_INVALID, "prefix runs need the fused stem");
    FRPCHK(plan_net(h, h->det, B, Hc, Wc, fused));
    StemParams sp{};
    sp.frames = h->det_scaled ? (const uint8_t*)h->scaled.p : (const uint8_t*)h->frames.p;
    sp.B = B; sp.H = h->dH; sp.W = h->dW;
    sp.row_stride = (long)h->dW * 3; sp.frame_stride = (long)h->dH * h->dW * 3;
    sp.Hc = Hc; sp.Wc = Wc; sp.Ho = Hc / 2; sp.Wo = Wc / 2;
    sp.rgb_in = 0;
    const bool det_wino = (long)B * (Hc / 8) * (Wc / 8) >= 2L * 240 * (h->n_cu > 0 ? h->n_cu : 256);
    double fl = 0.0;
    int64_t ln = 0;
    h->det_op_limit = n_ops;
    const int rc = run_net(h, h->det, B, Hc, Wc, &fl, &ln, &sp, nullptr, det_wino);
    h->det_op_limit = -1;
    if (rc != FRP_OK) return rc;
    const frp_conv_op& op = h->det.ops[n_ops - 1];
    const TensorDims d = h->det.dims[op.out_buf];
    if (th) *th = d.h;
    if (tw) *tw = d.w;
    if (tc) *tc = d.c;
    const int64_t need = (int64_t)B * d.h * d.w * d.c * 2;
    if (!out_f16) { HIPCHK(h, hipStreamSynchronize(h->stream)); return FRP_OK; }
    if (out_bytes < need) return fail(h, FRP_ERR_INVALID, "tensor buffer too small");
    HIPCHK(h, hipMemcpyAsync(out_f16, h->det.bufs[op.out_buf].p, (size_t)need, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return FRP_OK;
}

int frp_decode_heads(frp_handle* h, const void* head8, const void* head16, const void* head32, int32_t B, int32_t canvas_h,
                     int32_t canvas_w, int32_t max_faces, float det_thresh, float nms_iou, uint32_t flags, float* boxes,
                     float* kps, float* scores, int32_t* counts, int32_t* anchor_idx) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (!head8 || !head16 || !head32 || B <= 0 || B > 1024 || canvas_h <= 0 || canvas_w <= 0 || (canvas_h & 31) || (canvas_w & 31) ||
        max_faces <= 0 || max_faces > FRP_MAX_FACES_CAP)
        return fail(h, FRP_ERR_INVALID, "bad decode arguments");
    const void* src[3] = {head8, head16, head32};
    DevBuf tmp[3];
    DecodeParams dp{};
    int rc = FRP_OK;
    for (int l = 0; l < 3 && rc == FRP_OK; ++l) {
        dp.hl[l] = canvas_h / (8 << l);
        dp.wl[l] = canvas_w / (8 << l);
        const size_t bytes = (size_t)B * dp.hl[l] * dp.wl[l] * 32 * 2;
        rc = ensure(h, tmp[l], bytes);
        if (rc == FRP_OK && hipMemcpyAsync(tmp[l].p, src[l], bytes, hipMemcpyHostToDevice, h->stream) != hipSuccess)
            rc = fail(h, FRP_ERR_HIP, "head upload failed");
        dp.head[l] = (const _Float16*)tmp[l].p;
    }
    if (rc == FRP_OK) rc = ensure_results(h, B, max_faces);
    if (rc == FRP_OK) {
        dp.B = B; dp.max_faces = max_faces;
        const bool forced = flags & FRP_FLAG_FORCED_K;
        dp.logit_thresh = forced ? -INFINITY : logit_threshold(det_thresh);
        dp.nms_iou = forced ? 2.0f : nms_iou;
        dp.boxes = (float*)h->boxes.p; dp.kps = (float*)h->kps.p; dp.scores = (float*)h->scores.p;
        dp.anchor = (int32_t*)h->anchor.p; dp.counts = (int32_t*)h->counts.p;
        size_t anchors = 0;
        for (int l = 0; l < 3; ++l) anchors += (size_t)dp.hl[l] * dp.wl[l] * 2;
        rc = ensure(h, h->dense_logits, (size_t)B * anchors * 2);
        dp.logits = (_Float16*)h->dense_logits.p;
        hipError_t e = rc == FRP_OK ? launch_decode_nms(dp, h->stream) : hipSuccess;
        if (e != hipSuccess) rc = fail(h, FRP_ERR_HIP, std::string("decode_nms: ") + hipGetErrorString(e));
    }
    const size_t s = (size_t)B * max_faces;
    hipError_t e = hipSuccess;
    if (rc == FRP_OK) {
        if (boxes) e = hipMemcpyAsync(boxes, h->boxes.p, s * 16, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess && kps) e = hipMemcpyAsync(kps, h->kps.p, s * 40, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess && scores) e = hipMemcpyAsync(scores, h->scores.p, s * 4, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess && counts) e = hipMemcpyAsync(counts, h->counts.p, (size_t)B * 4, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess && anchor_idx) e = hipMemcpyAsync(anchor_idx, h->anchor.p, s * 4, hipMemcpyDeviceToHost, h->stream);
    }
    hipError_t e2 = hipStreamSynchronize(h->stream);
    for (int l = 0; l < 3; ++l) release(tmp[l]);
    if (rc != FRP_OK) return rc;
    if (e != hipSuccess || e2 != hipSuccess) return fail(h, FRP_ERR_HIP, "decode result copy failed");
    return FRP_OK;
}

// to_embedder: chips go to the embedder's input buffer (needs weights); else to h->scratch
static int align_common(frp_handle* h, const uint8_t* bgr, int H, int W, int64_t row_stride, const float* kps, int M, uint32_t flags,
                        bool to_embedder) {
    if (to_embedder && !h->have_weights) return fail(h, FRP_ERR_NO_WEIGHTS, "no weights loaded");
    if (!kps || M <= 0 || M > 65536) return fail(h, FRP_ERR_INVALID, "bad landmark arguments");
    FRPCHK(upload_frames(h, bgr, 1, H, W, row_stride));
    if (to_embedder) FRPCHK(plan_net(h, h->emb, M, FRP_CHIP, FRP_CHIP));
    else FRPCHK(ensure(h, h->scratch, (size_t)M * FRP_CHIP_PIX * 16));
    FRPCHK(ensure(h, h->kps, (size_t)M * 40));
    HIPCHK(h, hipMemcpyAsync(h->kps.p, kps, (size_t)M * 40, hipMemcpyHostToDevice, h->stream));
    AlignParams ap{};
    ap.frames = (const uint8_t*)h->frames.p;
    ap.B = 1; ap.H = H; ap.W = W;
    ap.row_stride = (long)W * 3; ap.frame_stride = (long)H * W * 3;
    ap.kps = (const float*)h->kps.p;
    ap.max_faces = M;
    ap.face_slot = nullptr;
    ap.n_faces = M;
    ap.rgb_in = (flags & FRP_FLAG_RGB) ? 1 : 0;
    ap.chips = to_embedder ? (_Float16*)h->emb.bufs[h->emb.in_buf].p : (_Float16*)h->scratch.p;
    hipError_t e = launch_align(ap, h->stream);
    if (e != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("align: ") + hipGetErrorString(e));
    return FRP_OK;
}

int frp_align(frp_handle* h, const uint8_t* bgr, int32_t H, int32_t W, int64_t row_stride, const float* kps, int32_t M,
              uint32_t flags, void* chips_f16) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (!chips_f16) return fail(h, FRP_ERR_INVALID, "null output");
    FRPCHK(align_common(h, bgr, H, W, row_stride, kps, M, flags, false));
    HIPCHK(h, hipMemcpyAsync(chips_f16, h->scratch.p, (size_t)M * FRP_CHIP_PIX * 8 * 2, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return FRP_OK;
}

int frp_embed_faces(frp_handle* h, const uint8_t* bgr, int32_t H, int32_t W, int64_t row_stride, const float* kps, int32_t M,
                    uint32_t flags, float* emb) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (!emb) return fail(h, FRP_ERR_INVALID, "null output");
    FRPCHK(align_common(h, bgr, H, W, row_stride, kps, M, flags, true));
    FRPCHK(run_embed(h, M));
    HIPCHK(h, hipMemcpyAsync(emb, h->emb.bufs[h->hdr.emb_out_buf].p, (size_t)M * FRP_EMB_DIM * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return FRP_OK;
}

int frp_embed_aligned(frp_handle* h, const uint8_t* chips, int32_t M, float* emb) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (!h->have_weights) return fail(h, FRP_ERR_NO_WEIGHTS, "no weights loaded");
    if (!chips || !emb || M <= 0 || M > 65536) return fail(h, FRP_ERR_INVALID, "bad chip arguments");
    FRPCHK(plan_net(h, h->emb, M, FRP_CHIP, FRP_CHIP));
    FRPCHK(ensure(h, h->scratch, (size_t)M * FRP_CHIP_PIX * 3));
    HIPCHK(h, hipMemcpyAsync(h->scratch.p, chips, (size_t)M * FRP_CHIP_PIX * 3, hipMemcpyHostToDevice, h->stream));
    hipError_t e = launch_chips_to_blob((const uint8_t*)h->scratch.p, M, (_Float16*)h->emb.bufs[h->emb.in_buf].p, h->stream);
    if (e != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("chips_to_blob: ") + hipGetErrorString(e));
    FRPCHK(run_embed(h, M));
    HIPCHK(h, hipMemcpyAsync(emb, h->emb.bufs[h->hdr.emb_out_buf].p, (size_t)M * FRP_EMB_DIM * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return FRP_OK;
}

static int match_common(frp_handle* h, const float* q, int M, float* all_scores_host, int32_t* idx, float* cos) {
    if (!q || M <= 0 || M > (1 << 20)) return fail(h, FRP_ERR_INVALID, "bad query arguments");
    if (h->g_rows <= 0) return fail(h, FRP_ERR_NO_GALLERY, "gallery is empty");
    const int mpad = round_up(M, 32);
    FRPCHK(ensure(h, h->q16, (size_t)mpad * FRP_EMB_DIM * 2));
    HIPCHK(h, hipMemsetAsync(h->q16.p, 0, (size_t)mpad * FRP_EMB_DIM * 2, h->stream));
    FRPCHK(upload_rows_normalized(h, q, M, (_Float16*)h->q16.p));
    DevBuf all;
    if (all_scores_host) FRPCHK(ensure(h, all, (size_t)M * h->g_rows * 4));
    int rc = run_match(h, M, (float*)all.p);
    hipError_t e = hipSuccess;
    if (rc == FRP_OK) {
        if (idx) e = hipMemcpyAsync(idx, h->best_idx.p, (size_t)M * 4, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess && cos) e = hipMemcpyAsync(cos, h->best_cos.p, (size_t)M * 4, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess && all_scores_host)
            e = hipMemcpyAsync(all_scores_host, all.p, (size_t)M * h->g_rows * 4, hipMemcpyDeviceToHost, h->stream);
    }
    hipError_t e2 = hipStreamSynchronize(h->stream);
    release(all);
    if (rc != FRP_OK) return rc;
    if (e != hipSuccess || e2 != hipSuccess) return fail(h, FRP_ERR_HIP, "match result copy failed");
    return FRP_OK;
}

int frp_match(frp_handle* h, const float* q, int32_t M, int32_t topk, int32_t* idx, float* cos) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (topk < 1 || topk > FRP_MAX_TOPK) return fail(h, FRP_ERR_INVALID, "topk must be in 1..FRP_MAX_TOPK");
    if (topk == 1) return match_common(h, q, M, nullptr, idx, cos);
    // k > 1: score matrix on the device (query chunks of <= 1 GiB of scores), then k selection passes per row
    if (!q || !idx || !cos || M <= 0 || M > (1 << 20)) return fail(h, FRP_ERR_INVALID, "bad query arguments");
    if (h->g_rows <= 0) return fail(h, FRP_ERR_NO_GALLERY, "gallery is empty");
    const long N = h->g_rows;
    int chunk = (int)std::max<long>(1, std::min<long>(M, (1L << 28) / N));
    DevBuf all, didx, dcos;
    int rc = ensure(h, all, (size_t)chunk * N * 4);
    if (rc == FRP_OK) rc = ensure(h, didx, (size_t)chunk * topk * 4);
    if (rc == FRP_OK) rc = ensure(h, dcos, (size_t)chunk * topk * 4);
    hipError_t e = hipSuccess;
    for (int m0 = 0; rc == FRP_OK && e == hipSuccess && m0 < M; m0 += chunk) {
        const int m = std::min(chunk, M - m0);
        const int mpad = round_up(m, 32);
        rc = ensure(h, h->q16, (size_t)mpad * FRP_EMB_DIM * 2);
        if (rc != FRP_OK) break;
        e = hipMemsetAsync(h->q16.p, 0, (size_t)mpad * FRP_EMB_DIM * 2, h->stream);
        if (e != hipSuccess) break;
        rc = upload_rows_normalized(h, q + (size_t)m0 * FRP_EMB_DIM, m, (_Float16*)h->q16.p);
        if (rc == FRP_OK) rc = run_match(h, m, (float*)all.p);
        if (rc != FRP_OK) break;
        e = launch_topk_rows((const float*)all.p, m, N, topk, (int32_t*)didx.p, (float*)dcos.p, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(idx + (size_t)m0 * topk, didx.p, (size_t)m * topk * 4, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(cos + (size_t)m0 * topk, dcos.p, (size_t)m * topk * 4, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    }
    (void)hipStreamSynchronize(h->stream);
    release(all); release(didx); release(dcos);
    if (rc != FRP_OK) return rc;
    if (e != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("match top-k: ") + hipGetErrorString(e));
    return FRP_OK;
}

int frp_match_scores(frp_handle* h, const float* q, int32_t M, float* cos_all, int64_t n_cols) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (!cos_all) return fail(h, FRP_ERR_INVALID, "null output");
    // cos_all holds M x n_cols floats: the gallery may have grown since the caller read its size
    if (n_cols != h->g_rows) return fail(h, FRP_ERR_INVALID, "match_scores: output sized for another gallery size");
    return match_common(h, q, M, cos_all, nullptr, nullptr);
}

int frp_conv2d_nhwc(frp_handle* h, const void* x, int32_t N, int32_t H, int32_t W, int32_t Cin, const void* w, int32_t Cout,
                    int32_t ksize, int32_t stride, const float* bias, const float* slope, const void* res, int32_t res_h,
                    int32_t res_w, int32_t act, int32_t flags, void* out) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (!x || !w || !bias || !out || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || !(ksize == 1 || ksize == 3) ||
        !(stride == 1 || stride == 2))
        return fail(h, FRP_ERR_INVALID, "bad conv arguments");
    const int pad = ksize / 2;
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    const bool up2 = flags & FRP_FLAG_RES_UP2;
    const size_t xb = (size_t)N * H * W * Cin * 2, wb = (size_t)Cout * ksize * ksize * Cin * 2;
    const size_t bb = (size_t)Cout * 4 * ((flags & FRP_FLAG_BORDER_BIAS) ? 9 : 1);
    const size_t ob = (size_t)N * Ho * Wo * Cout * ((flags & FRP_FLAG_OUT_F32) ? 4 : 2);
    const size_t rb = res ? (size_t)N * (up2 ? res_h : Ho) * (up2 ? res_w : Wo) * Cout * 2 : 0;
    DevBuf dx, dw, db, ds, dr, dout, dwino;
    int rc = ensure(h, dx, xb);
    if (rc == FRP_OK) rc = ensure(h, dw, wb);
    if (rc == FRP_OK) rc = ensure(h, db, bb);
    if (rc == FRP_OK) rc = ensure(h, dout, ob);
    if (rc == FRP_OK && slope) rc = ensure(h, ds, (size_t)Cout * 4);
    if (rc == FRP_OK && res) rc = ensure(h, dr, rb);
    // flags bit 16: through the Winograd kernel (parity tests); an ineligible shape is an error, not a silent fallback
    const bool want_wino = (flags & 0x10000) != 0;
    std::vector<uint16_t> wimg;
    if (want_wino) {
        bool shape_ok = conv3x3_wino_shape_ok(W, Cin, ksize, stride) ||
                        (ksize == 3 && stride == 1 && conv3x3_wino_wide_pays(N, H, W, Cin, Cout, h->n_cu, res != nullptr));      // (2-D tiles: wide maps)
#ifdef FRP_LAB
        shape_ok = shape_ok || (((((flags >> 8) & 0xff) & 64) || (flags & 0x80000)) && conv3x3_wino_lab_shape_ok(W, Cin, ksize, stride));
#endif
        if (!shape_ok || (flags & (FRP_FLAG_OUT_F32 | FRP_FLAG_RES_UP2)))
            return fail(h, FRP_ERR_INVALID, "shape not covered by the Winograd kernel");
        wimg.resize(conv3x3_wino_image_bytes(Cin, Cout) / 2);
        build_wino_image((const uint16_t*)w, Cin, Cout, wimg.data());
        if (rc == FRP_OK) rc = ensure(h, dwino, wimg.size() * 2);
    }
    hipError_t e = hipSuccess;
    if (rc == FRP_OK) {
        e = hipMemcpyAsync(dx.p, x, xb, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess && want_wino) e = hipMemcpyAsync(dwino.p, wimg.data(), wimg.size() * 2, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dw.p, w, wb, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(db.p, bias, bb, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess && slope) e = hipMemcpyAsync(ds.p, slope, (size_t)Cout * 4, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess && res) e = hipMemcpyAsync(dr.p, res, rb, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) {
            ConvParams p{};
            p.x = (const _Float16*)dx.p; p.w = (const _Float16*)dw.p; p.bias = (const float*)db.p;
            p.slope = slope ? (const float*)ds.p : nullptr;
            p.res = res ? (const _Float16*)dr.p : nullptr;
            p.out = dout.p;
            p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KS = ksize; p.stride = stride; p.act = act;
            p.flags = flags & (FRP_FLAG_BORDER_BIAS | FRP_FLAG_OUT_F32 | FRP_FLAG_RES_UP2);
            p.dbg = ((flags >> 8) & 0xff) | ((flags & 0x80000) ? 256 : 0) | ((flags & 0x100000) ? 512 : 0) | ((flags & 0x200000) ? 2048 : 0) | ((flags & 0x400000) ? 4096 : 0);      // (bit 20: not the 64 -> 64 kernel; bit 21: the opt-in stride-2 row-patch kernel)      // kernel A/B switches (tests: 1 = generic kernel instead of the
                                                                                // row-patch one; flags bit 19 = dbg 256: the Winograd kernel's 2-D tiles, lab build)
            // flags bit 17 / 18: quarter tiles always / never (default: by the tile count, conv_common.h: conv_small_m)
            p.small_m = (flags & 0x20000) ? 1 : ((flags & 0x40000) || want_wino) ? -1 : 0;
            p.Hr = res_h; p.Wr = res_w;
            if (want_wino) p.wino_w = (const _Float16*)dwino.p;
            e = launch_conv(p, h->stream);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(out, dout.p, ob, hipMemcpyDeviceToHost, h->stream);
    }
    hipError_t e2 = hipStreamSynchronize(h->stream);
    DevBuf* all[] = {&dx, &dw, &db, &ds, &dr, &dout, &dwino};
    for (DevBuf* b : all) release(*b);
    if (rc != FRP_OK) return rc;
    if (e != hipSuccess) return fail(h, e == hipErrorInvalidValue ? FRP_ERR_INVALID : FRP_ERR_HIP, std::string("conv2d: ") + hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("conv2d sync: ") + hipGetErrorString(e2));
    return FRP_OK;
}

int frp_conv2d_f8(frp_handle* h, const void* x8, int32_t N, int32_t H, int32_t W, int32_t Cin, const void* w8, int32_t Cout,
                  const float* wscale, const float* bias, const float* slope, const void* res16, int32_t act, int32_t flags,
                  float in_scale, float out_scale, void* out, void* out2_f8) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (!x8 || !w8 || !wscale || !bias || !out || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0)
        return fail(h, FRP_ERR_INVALID, "bad conv arguments");
    const bool out8 = (flags & FRP_FLAG_OUT_FP8) != 0;
    const size_t xb = (size_t)N * H * W * Cin, wb = (size_t)Cout * 9 * Cin, on = (size_t)N * H * W * Cout;
    const size_t bb = (size_t)Cout * 4 * ((flags & FRP_FLAG_BORDER_BIAS) ? 9 : 1);
    DevBuf dx, dw, dws, db, ds, dr, dout, dout2;
    int rc = ensure(h, dx, xb);
    if (rc == FRP_OK) rc = ensure(h, dw, wb);
    if (rc == FRP_OK) rc = ensure(h, dws, (size_t)Cout * 4);
    if (rc == FRP_OK) rc = ensure(h, db, bb);
    if (rc == FRP_OK) rc = ensure(h, dout, on * (out8 ? 1 : 2));
    if (rc == FRP_OK && out2_f8) rc = ensure(h, dout2, on);
    if (rc == FRP_OK && slope) rc = ensure(h, ds, (size_t)Cout * 4);
    if (rc == FRP_OK && res16) rc = ensure(h, dr, on * 2);
    hipError_t e = hipSuccess;
    if (rc == FRP_OK) {
        e = hipMemcpyAsync(dx.p, x8, xb, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dw.p, w8, wb, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dws.p, wscale, (size_t)Cout * 4, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(db.p, bias, bb, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess && slope) e = hipMemcpyAsync(ds.p, slope, (size_t)Cout * 4, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess && res16) e = hipMemcpyAsync(dr.p, res16, on * 2, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) {
            ConvParams p{};
            p.x = (const _Float16*)dx.p; p.w = (const _Float16*)dw.p; p.bias = (const float*)db.p;
            p.slope = slope ? (const float*)ds.p : nullptr;
            p.res = res16 ? (const _Float16*)dr.p : nullptr;
            p.out = dout.p;
            p.out2 = out2_f8 ? dout2.p : nullptr;
            p.wscale = (const float*)dws.p;
            p.in_scale = in_scale; p.out_scale = out_scale;
            p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KS = 3; p.stride = 1; p.act = act;
            p.flags = (flags & (FRP_FLAG_BORDER_BIAS | FRP_FLAG_OUT_FP8)) | FRP_FLAG_F8;
            e = launch_conv(p, h->stream);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(out, dout.p, on * (out8 ? 1 : 2), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess && out2_f8) e = hipMemcpyAsync(out2_f8, dout2.p, on, hipMemcpyDeviceToHost, h->stream);
    }
    hipError_t e2 = hipStreamSynchronize(h->stream);
    DevBuf* all[] = {&dx, &dw, &dws, &db, &ds, &dr, &dout, &dout2};
    for (DevBuf* b : all) release(*b);
    if (rc != FRP_OK) return rc;
    if (e != hipSuccess) return fail(h, e == hipErrorInvalidValue ? FRP_ERR_INVALID : FRP_ERR_HIP, std::string("conv2d_f8: ") + hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("conv2d_f8 sync: ") + hipGetErrorString(e2));
    return FRP_OK;
}

#ifdef FRP_LAB   // tuning hooks (include/frp_lab.h): only in libfrp_lab.so
int frp_conv_bench(frp_handle* h, int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize, int32_t stride,
                   int32_t act, int32_t flags, int32_t with_res, int32_t iters, float* ms_avg, uint64_t* stamps_out) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (!ms_avg || iters <= 0 || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || !(ksize == 1 || ksize == 3) ||
        !(stride == 1 || stride == 2))
        return fail(h, FRP_ERR_INVALID, "bad bench arguments");
    const int pad = ksize / 2;
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    const size_t xn = (size_t)N * H * W * Cin, wn = (size_t)Cout * ksize * ksize * Cin, on = (size_t)N * Ho * Wo * Cout;
    const bool f8 = (flags & FRP_FLAG_F8) != 0;          // fp8 operands: random fp16 bit patterns read as E4M3 bytes (timing only)
    DevBuf dx, dw, db, ds, dr, dout;
    int rc = ensure(h, dx, xn * 2);
    if (rc == FRP_OK) rc = ensure(h, dw, wn * 2);
    if (rc == FRP_OK) rc = ensure(h, db, (size_t)Cout * 4 * 9);
    if (rc == FRP_OK) rc = ensure(h, ds, (size_t)Cout * 4);
    if (rc == FRP_OK) rc = ensure(h, dr, on * 2);
    if (rc == FRP_OK) rc = ensure(h, dout, on * 4);
    hipError_t e = hipSuccess;
    float ms = 0.f;
    if (rc == FRP_OK) {
        e = launch_fill_random_f16((_Float16*)dx.p, (long)xn, 1u, 1.0f, h->stream);
        if (e == hipSuccess) e = launch_fill_random_f16((_Float16*)dw.p, (long)wn, 2u, 1.0f / sqrtf((float)(ksize * ksize * Cin)), h->stream);
        if (e == hipSuccess) e = launch_fill_random_f16((_Float16*)dr.p, (long)on, 3u, 1.0f, h->stream);
        if (e == hipSuccess) e = hipMemsetAsync(db.p, 0, (size_t)Cout * 36, h->stream);
        if (e == hipSuccess) e = hipMemsetAsync(ds.p, 0, (size_t)Cout * 4, h->stream);
        ConvParams p{};
        p.x = (const _Float16*)dx.p; p.w = (const _Float16*)dw.p; p.bias = (const float*)db.p; p.slope = (const float*)ds.p;
        p.res = with_res ? (const _Float16*)dr.p : nullptr; p.out = dout.p;
        p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KS = ksize; p.stride = stride; p.act = act;
        p.flags = flags & (FRP_FLAG_BORDER_BIAS | FRP_FLAG_OUT_F32 | FRP_FLAG_F8 | FRP_FLAG_OUT_FP8);
        if (f8) {                                         // unit scales in the slope buffer's neighbour: reuse the bias buffer (zeros) + 1
            p.wscale = (const float*)ds.p;                // zeros: products vanish, timing is unaffected
            p.in_scale = p.out_scale = 1.0f;
            if (!(flags & FRP_FLAG_OUT_FP8)) p.out2 = dr.p;   // conv2-style: fp16 out + fp8 copy (residual buffer doubles as the copy target when unused)
            if (with_res) p.out2 = nullptr;
        }
        p.dbg = ((flags >> 8) & 0xff) | ((flags & 0x80000) ? 256 : 0) | ((flags & 0x100000) ? 512 : 0) | ((flags & 0x200000) ? 2048 : 0) | ((flags & 0x400000) ? 4096 : 0);      // (bit 20: not the 64 -> 64 kernel)
        p.small_m = (flags & 0x20000) ? 1 : ((flags & 0x40000) || (flags & 0x10000)) ? -1 : 0;
        DevBuf dwino;
        bool wino_shape = conv3x3_wino_shape_ok(W, Cin, ksize, stride) || (ksize == 3 && stride == 1 && conv3x3_wino_wide_pays(N, H, W, Cin, Cout, h->n_cu, with_res != 0));
#ifdef FRP_LAB
        wino_shape = wino_shape || ((p.dbg & (64 | 256)) && conv3x3_wino_lab_shape_ok(W, Cin, ksize, stride));
#endif
        if ((flags & 0x10000) && wino_shape) {     // Winograd kernel: a random weight image (timing only)
            const size_t ib = conv3x3_wino_image_bytes(Cin, Cout);
            if (ensure(h, dwino, ib) == FRP_OK) {
                e = launch_fill_random_f16((_Float16*)dwino.p, (long)(ib / 2), 5u, 1.0f / sqrtf((float)(9 * Cin)), h->stream);
                p.wino_w = (const _Float16*)dwino.p;
            }
        }
        DevBuf dst;
        if (stamps_out && ensure(h, dst, 256 * 8 * 8) == FRP_OK) {
            (void)hipMemsetAsync(dst.p, 0, 256 * 8 * 8, h->stream);
            p.stamps = (unsigned long long*)dst.p;
        }
        for (int i = 0; i < 2 && e == hipSuccess; ++i) e = launch_conv(p, h->stream);   // warm-up
        if (e == hipSuccess) e = hipEventRecord(h->ev[0], h->stream);
        for (int i = 0; i < iters && e == hipSuccess; ++i) e = launch_conv(p, h->stream);
        if (e == hipSuccess) e = hipEventRecord(h->ev[1], h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, h->ev[0], h->ev[1]);
        if (e == hipSuccess && stamps_out && p.stamps) e = hipMemcpy(stamps_out, p.stamps, 256 * 8 * 8, hipMemcpyDeviceToHost);
        release(dst);
        (void)hipStreamSynchronize(h->stream);
        release(dwino);
    }
    (void)hipStreamSynchronize(h->stream);
    DevBuf* all[] = {&dx, &dw, &db, &ds, &dr, &dout};
    for (DevBuf* b : all) release(*b);
    if (rc != FRP_OK) return rc;
    if (e != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("conv_bench: ") + hipGetErrorString(e));
    *ms_avg = ms / iters;
    return FRP_OK;
}

int frp_mfma_peak(frp_handle* h, int32_t waves_per_simd, int32_t iters, float* tflops) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    // waves_per_simd 1..8: register-operand loop.  16*r + 2 (r = 4, 3, 2): the conv k-step mix - 8 waves per CU,
    // r ds_read_b128 per 4 MFMAs - one 512-thread block per CU
    const int lds_reads = waves_per_simd >> 4;
    if (lds_reads) waves_per_simd &= 15;
    if (!tflops || iters <= 0 || waves_per_simd < 1 || waves_per_simd > 8 || (lds_reads && (waves_per_simd != 2 || lds_reads < 2 || lds_reads > 4)))
        return fail(h, FRP_ERR_INVALID, "bad arguments");
    const int blocks = lds_reads ? h->n_cu : 256 * waves_per_simd;   // 256 CUs x (4 waves per block = one per SIMD)
    DevBuf src, dst;
    int rc = ensure(h, src, 3 * 384 * 128);
    if (rc == FRP_OK) rc = ensure(h, dst, (size_t)blocks * 512 * 4);
    hipError_t e = hipSuccess;
    float ms = 0.f;
    if (rc == FRP_OK) {
        e = launch_fill_random_f16((_Float16*)src.p, 3 * 384 * 64, 7u, 1.0f, h->stream);
        auto run = [&]() {
            return lds_reads ? launch_mfma_lds((const _Float16*)src.p, (float*)dst.p, blocks, lds_reads, iters, h->stream)
                             : launch_mfma_peak((const _Float16*)src.p, (float*)dst.p, blocks, iters, h->stream);
        };
        if (e == hipSuccess) e = run();
        if (e == hipSuccess) e = hipEventRecord(h->ev[0], h->stream);
        if (e == hipSuccess) e = run();
        if (e == hipSuccess) e = hipEventRecord(h->ev[1], h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, h->ev[0], h->ev[1]);
    }
    (void)hipStreamSynchronize(h->stream);
    release(src);
    release(dst);
    if (rc != FRP_OK) return rc;
    if (e != hipSuccess) return fail(h, FRP_ERR_HIP, std::string("mfma_peak: ") + hipGetErrorString(e));
    *tflops = (float)((double)blocks * (lds_reads ? 8.0 * 16 : 4.0 * 4) * iters * 32768.0 / (ms * 1e-3) / 1e12);
    return FRP_OK;
}

int frp_kstep_lab(frp_handle* h, int32_t variant, int32_t iters, float* tflops) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    if (!tflops || iters <= 0) return fail(h, FRP_ERR_INVALID, "bad arguments");
    const int blocks = h->n_cu;
    DevBuf src, dst;
    int rc = ensure(h, src, 4u << 20);        // LDS image + the 4 MiB window the lab's LDS-DMA variants read
    if (rc == FRP_OK) rc = ensure(h, dst, (size_t)blocks * 512 * 4);
    hipError_t e = hipSuccess;
    float ms = 0.f;
    if (rc == FRP_OK) {
        e = launch_fill_random_f16((_Float16*)src.p, 2L << 20, 7u, 1.0f, h->stream);
        if (e == hipSuccess) e = launch_kstep_lab((const _Float16*)src.p, (float*)dst.p, blocks, variant, iters, h->stream);
        if (e == hipSuccess) e = hipEventRecord(h->ev[0], h->stream);
        if (e == hipSuccess) e = launch_kstep_lab((const _Float16*)src.p, (float*)dst.p, blocks, variant, iters, h->stream);
        if (e == hipSuccess) e = hipEventRecord(h->ev[1], h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, h->ev[0], h->ev[1]);
    }
    (void)hipStreamSynchronize(h->stream);
    release(src);
    release(dst);
    if (rc != FRP_OK) return rc;
    if (e != hipSuccess) return fail(h, e == hipErrorInvalidValue ? FRP_ERR_INVALID : FRP_ERR_HIP, std::string("kstep_lab: ") + hipGetErrorString(e));
    // fp8 variants (bit 10): 8 MFMAs of 32x32x64 per wave and step = twice the FLOPs of the fp16 step
    *tflops = (float)((double)blocks * kstep_lab_waves(variant) * 16 * kstep_lab_steps_per_iter(variant) * iters * 32768.0 * ((variant & 1024) ? 2.0 : 1.0) /
                      (ms * 1e-3) / 1e12);
    return FRP_OK;
}

#endif  // FRP_LAB

int frp_get_counters(frp_handle* h, frp_counters* out) {
    if (!h || !out) return FRP_ERR_INVALID;
    Guard g(h);
    h->ctr.struct_size = sizeof(frp_counters);
    h->ctr.gallery_rows = h->g_rows;
    *out = h->ctr;
    return FRP_OK;
}

int frp_set_profile(frp_handle* h, int32_t on) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    h->cfg.profile = on ? 1 : 0;
    return FRP_OK;
}

int frp_reset_counters(frp_handle* h) {
    if (!h) return FRP_ERR_INVALID;
    Guard g(h);
    h->ctr = frp_counters{};
    h->ctr.struct_size = sizeof(frp_counters);
    return FRP_OK;
}

}  // extern "C"
