// api.cpp -- the C ABI of include/gtcrn_micro_hip.h: handle, workspace, launch sequencing.
// Host code only; every arithmetic step runs in kernels.hip on the GPU.  There is no CPU fallback:
// without a gfx950 device gtcrn_model_create fails.
#include "../../include/gtcrn_micro_hip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "kernels.h"
#include "layout.h"
#include "pack.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
int hip_fail(hipError_t e, const char* what) {
    return fail(GTCRN_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr)                                         \
    do {                                                      \
        hipError_t e_ = (expr);                               \
        if (e_ != hipSuccess) return hip_fail(e_, #expr);     \
    } while (0)
#define LAUNCH_TRY(expr)                                                         \
    do {                                                                         \
        int e_ = (expr);                                                         \
        if (e_ != 0) return hip_fail(static_cast<hipError_t>(e_), #expr);        \
    } while (0)

// HIP-event timing of every kernel launch, kept per call so that a whole timed region can be
// averaged afterwards without synchronising inside it.
struct Timing {
    int kernel = 0;  // index into kKernelNames
    hipEvent_t a = nullptr, b = nullptr;
};
// every timed launch records WHICH kernel it was (mixed offline / streaming calls keep their own rows)
enum KernelId { K_STFT, K_ENCODER, K_GTCN1, K_GTCN2, K_DECODER, K_ISTFT, K_FRONT, K_ENCODER_GT, K_GTCN_MS, K_STREAM_MS, K_STREAM_WIDE, K_COUNT };
const char* const kKernelNames[K_COUNT] = {"k_stft",  "k_encoder",    "k_gtcn1",   "k_gtcn2",    "k_decoder",
                                           "k_istft", "k_front",      "k_encoder_gt", "k_gtcn_ms", "k_stream_ms",
                                           "k_stream_wide"};
constexpr int kNumKernels = K_COUNT;

}  // namespace

// train.cpp reports its errors through the same thread-local message
extern "C" void gtcrn_set_error_(const char* msg) { g_err = msg ? msg : ""; }

struct gtcrn_model {
    int device = 0;
    float* d_pf = nullptr;   // packed floats (gtl::P_FLOATS)
    float* d_pfq = nullptr;  // the same with int8-quantised weights (BASELINE configs[4] variant)
    int* d_pi = nullptr;     // packed ints (gtl::P_INTS)
    float* d_twid = nullptr; // 512 complex twiddles
    int* d_pref = nullptr;   // prefix table of a variable-length batch (1025 ints, see gtk::launch_len_prefix)
    bool var_spans = true;   // variable-length batches run in time spans (off: one workgroup per utterance, the A/B switch)
    int stream_form = 0;     // single-frame streaming steps: 0 = ONE launch, the form picked by the stream count (four
                             // streams per workgroup, k_stream_ms, or seven, k_stream_wide: stream_wide_pays); 1 = the
                             // three-launch form (encoder / both GTCN stacks / decoder, hand-offs through HBM); 2 / 3 =
                             // ONE launch, k_stream_ms / k_stream_wide whatever the count: the A/B switches
    std::vector<int> h_pi;   // host copy of the int tables (slot permutations for the debug taps)
    // workspace for B x T
    long cap_bt = 0;         // capacity in (batch * frames)
    int last_B = 0, last_T = 0;
    float* d_en0 = nullptr;  // (B,T,65,16)
    float* d_en[4] = {nullptr, nullptr, nullptr, nullptr};  // en1..en4 (B,T,33,16)
    float* d_g1 = nullptr;   // gtcn1 output
    float* d_g2 = nullptr;   // gtcn2 output
    float* d_spec_a = nullptr;  // frame-major spectrograms for forward_wave (B,T,257,2)
    float* d_spec_b = nullptr;
    bool debug = false;
    bool debug_keep_fused = false;   // gtcrn_debug_enable(m, 2): phase stamps only -- single-frame steps stay ONE launch
    float* d_dbg = nullptr;
    long dbg_cap_bt = 0;
    unsigned long long* d_stamps = nullptr;  // [4 kernels][B][16] phase cycle sums (diagnostic build only)
    int stamps_cap_b = 0;
    float** d_ptr8 = nullptr;  // device table of 8 tcn cache pointers
    bool last_quant = false;       // the last forward was the fp16 variant: the hand-off buffers hold fp16 records
    bool last_fused_stream = false;  // the last forward was the single-launch streaming step: no hand-off tensors exist
    bool timing = false;
    int timing_only = -1;          // >= 0: record events around this kernel only (two events per call)
    std::vector<Timing> timings;   // one entry per timed launch since gtcrn_timing_enable(m, 1)
    std::vector<Timing> ev_pool;   // recycled events
};

namespace {

void free_workspace(gtcrn_model* m) {
    float** bufs[] = {&m->d_en0, &m->d_en[0], &m->d_en[1], &m->d_en[2], &m->d_en[3], &m->d_g1, &m->d_g2,
                      &m->d_spec_a, &m->d_spec_b};
    for (float** p : bufs) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    m->cap_bt = 0;
}

int ensure_workspace(gtcrn_model* m, int B, int T, hipStream_t s) {
    const long bt = (long)B * T;
    if (bt > m->cap_bt) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return fail(GTCRN_ERR_STATE, "workspace too small during stream capture: call gtcrn_model_reserve first");
        HIP_TRY(hipDeviceSynchronize());
        free_workspace(m);
        HIP_TRY(hipMalloc(&m->d_en0, sizeof(float) * bt * 65 * 16));
        for (int i = 0; i < 4; ++i) HIP_TRY(hipMalloc(&m->d_en[i], sizeof(float) * bt * 528));
        HIP_TRY(hipMalloc(&m->d_g1, sizeof(float) * bt * 528));
        HIP_TRY(hipMalloc(&m->d_g2, sizeof(float) * bt * 528));
        HIP_TRY(hipMalloc(&m->d_spec_a, sizeof(float) * bt * 514));
        HIP_TRY(hipMalloc(&m->d_spec_b, sizeof(float) * bt * 514));
        m->cap_bt = bt;
    }
    if (m->debug && bt > m->dbg_cap_bt) {
        if (m->d_dbg) (void)hipFree(m->d_dbg);
        m->d_dbg = nullptr;
        HIP_TRY(hipMalloc(&m->d_dbg, sizeof(float) * bt * (3 * 528 + 65 * 16 + 2 * 129)));
        m->dbg_cap_bt = bt;
    }
    if (m->debug && B + 2048 > m->stamps_cap_b) {
        // one row per WORKGROUP: launches in time spans run up to 2048 of them whatever the batch
        const int rows = B + 2048;        // (a split plan: nA <= B one-per-utterance rows, then up to 2048 share rows behind them)
        if (m->d_stamps) (void)hipFree(m->d_stamps);
        m->d_stamps = nullptr;
        HIP_TRY(hipMalloc(&m->d_stamps, sizeof(unsigned long long) * 4 * rows * 16));
        HIP_TRY(hipMemset(m->d_stamps, 0, sizeof(unsigned long long) * 4 * rows * 16));
        m->stamps_cap_b = rows;
    }
    return 0;
}

struct Timer {
    gtcrn_model* m;
    hipStream_t s;
    Timer(gtcrn_model* m_, hipStream_t s_) : m(m_), s(s_) {}
    bool on = false;
    void begin(int kernel) {
        on = m->timing && (m->timing_only < 0 || m->timing_only == kernel);
        if (!on) return;
        Timing t;
        if (!m->ev_pool.empty()) {
            t = m->ev_pool.back();
            m->ev_pool.pop_back();
        } else {
            (void)hipEventCreate(&t.a);
            (void)hipEventCreate(&t.b);
        }
        t.kernel = kernel;
        (void)hipEventRecord(t.a, s);
        m->timings.push_back(t);
    }
    void end() {
        if (!on) return;
        (void)hipEventRecord(m->timings.back().b, s);
    }
};

// Single-frame steps of B streams: k_stream_ms serves four streams per workgroup, k_stream_wide seven in a workgroup that
// takes WIDE_COST times as long (measured: profiles/r06_ab_stream_wide.txt).  Both run one workgroup per CU, so a step
// costs rounds-of-256-workgroups x the form's time: the wide form pays once the narrow one needs more rounds than it.
static bool stream_wide_pays(int B) {
    constexpr double WIDE_COST = 1.5;      // 49.3 us against 32.9 us per round of 256 workgroups at 65 536 streams
    const int ws = gtk::stream_wide_streams();
    const long r4 = ((B + 3) / 4 + 255) / 256, r7 = ((B + ws - 1) / ws + 255) / 256;
    return (double)r7 * WIDE_COST < (double)r4;
}

// the five model kernels on one stream; state == nullptr for offline
int run_model(gtcrn_model* m, const float* spec_in, long isb, long isf, long ist, float* spec_out, long osb, long osf,
              long ost, int B, int T, float* state, hipStream_t s, const int* lens = nullptr,
              const gtk::Quant* q = nullptr, bool front_done = false) {
    Timer tm(m, s);
    const float* pf = q ? m->d_pfq : m->d_pf;
    const bool offline = !state;
    m->last_quant = q != nullptr;
    m->last_fused_stream = false;
    // single-frame streaming step: ONE launch, nothing handed over through HBM (the three-launch form below remains
    // for the stage taps of the parity tests, which read the hand-off tensors)
    if (state && T == 1 && !q && (!m->debug || m->debug_keep_fused) && m->stream_form != 1 && gtk::stream_ms_usable(isb, osb)) {
        const bool wide = m->stream_form == 3 || (m->stream_form == 0 && stream_wide_pays(B));
        unsigned long long* stamps = (m->debug && m->d_stamps) ? m->d_stamps : nullptr;
        tm.begin(wide ? K_STREAM_WIDE : K_STREAM_MS);
        if (wide) LAUNCH_TRY(gtk::launch_stream_wide(spec_in, isb, isf, spec_out, osb, osf, B, pf, m->d_pi, state, stamps, s));
        else LAUNCH_TRY(gtk::launch_stream_ms(spec_in, isb, isf, spec_out, osb, osf, B, pf, m->d_pi, state, stamps, s));
        tm.end();
        m->last_fused_stream = true;
        m->last_B = B;
        m->last_T = T;
        return 0;
    }
    if (offline) {
        // offline: the frame-independent front end runs as a throughput kernel (fused with the STFT by
        // forward_wave_impl, which passes front_done), the per-utterance kernel keeps the three GTConv blocks
        if (!front_done) {
            tm.begin(K_FRONT);
            LAUNCH_TRY(gtk::launch_front(nullptr, 0, spec_in, isb, isf, ist, B, T, lens, nullptr, nullptr, pf, m->d_pi,
                                         nullptr, m->d_en0, m->d_en[0], s, q));
            tm.end();
        }
    }
    // variable-length offline batches: the per-utterance kernels share the frames that exist (time spans over a prefix
    // table of the lengths) -- unless the batch is whole rounds of the chip already
    const int* pref = nullptr;
    if (offline && lens && !q && front_done && m->var_spans && gtk::var_spans_usable(B) && B % 256 != 0) {
        LAUNCH_TRY(gtk::launch_len_prefix(lens, B, T, m->d_pref, s));
        pref = m->d_pref;
    }
    unsigned long long* stp = (m->debug && m->d_stamps) ? m->d_stamps : nullptr;
    const long sst = (long)m->stamps_cap_b * 16;
    tm.begin(offline ? K_ENCODER_GT : K_ENCODER);
    LAUNCH_TRY(gtk::launch_encoder(spec_in, isb, isf, ist, B, T, lens, pf, m->d_pi, m->d_en0, m->d_en[0], m->d_en[1],
                                   m->d_en[2], m->d_en[3], state, stp ? stp : nullptr, s, q, offline, pref));
    tm.end();
    // GTCN: offline calls (no stream state) use the frequency-band form (registers + wave-private LDS, no barrier);
    // single-frame streaming steps run BOTH stacks per position in one launch; other streaming chunkings use the
    // ring form, whose chunks may hold any number of frames
    if (state && T == 1) {
        tm.begin(K_GTCN_MS);
        LAUNCH_TRY(gtk::launch_gtcn_ms(m->d_en[3], m->d_g1, m->d_g2, pf + gtl::P_GTCN, B, state, s));
        tm.end();
    } else {
        tm.begin(K_GTCN1);
        if (!state)
            LAUNCH_TRY(gtk::launch_gtcn_band(m->d_en[3], m->d_g1, pf + gtl::P_GTCN, B, T, lens, nullptr, s, q, pref));
        else
            LAUNCH_TRY(gtk::launch_gtcn(m->d_en[3], m->d_g1, pf + gtl::P_GTCN, B, T, state, gtk::ST_G1_H, nullptr,
                                        stp ? stp + sst : nullptr, s));
        tm.end();
        tm.begin(K_GTCN2);
        // the second stack stores gtcn2(x) + en_outs[4]: exactly the decoder's first input (Decoder.forward :467)
        if (!state)
            LAUNCH_TRY(gtk::launch_gtcn_band(m->d_g1, m->d_g2, pf + gtl::P_GTCN + gtl::GTCN_SIZE, B, T, lens, m->d_en[3], s,
                                             q, pref));
        else
            LAUNCH_TRY(gtk::launch_gtcn(m->d_g1, m->d_g2, pf + gtl::P_GTCN + gtl::GTCN_SIZE, B, T, state, gtk::ST_G2_H,
                                        m->d_en[3], stp ? stp + 2 * sst : nullptr, s));
        tm.end();
    }
    tm.begin(K_DECODER);
    LAUNCH_TRY(gtk::launch_decoder(m->d_g2, m->d_en0, m->d_en[0], m->d_en[1], m->d_en[2], m->d_en[3], spec_in, isb, isf,
                                   ist, spec_out, osb, osf, ost, B, T, lens, pf, m->d_pi, state,
                                   m->debug && !q ? m->d_dbg : nullptr, stp ? stp + 3 * sst : nullptr, s, q, pref));
    tm.end();
    m->last_B = B;
    m->last_T = T;
    return 0;
}

int check_model(gtcrn_model* m) {
    if (!m || !m->d_pf) return fail(GTCRN_ERR_STATE, "invalid model handle");
    HIP_TRY(hipSetDevice(m->device));
    return 0;
}

int upload_params(gtcrn_model* m, const float* h_params, long n) {
    std::vector<float> F(gtl::P_FLOATS);
    std::vector<int> I(gtl::P_INTS);
    std::string err;
    if (gtcrn::pack_params(h_params, n, F.data(), I.data(), err) != 0) return fail(GTCRN_ERR_ARG, err);
    HIP_TRY(hipMemcpy(m->d_pf, F.data(), sizeof(float) * gtl::P_FLOATS, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(m->d_pi, I.data(), sizeof(int) * gtl::P_INTS, hipMemcpyHostToDevice));
    gtcrn::quantize_packed(F.data());
    HIP_TRY(hipMemcpy(m->d_pfq, F.data(), sizeof(float) * gtl::P_FLOATS, hipMemcpyHostToDevice));
    m->h_pi = I;
    return 0;
}

}  // namespace

extern "C" {

int gtcrn_abi_version(void) { return GTCRN_ABI_VERSION; }
const char* gtcrn_last_error(void) { return g_err.c_str(); }

long gtcrn_param_tensors(void) { return (long)gtcrn::param_table().size(); }
const char* gtcrn_param_name(long i) {
    const auto& t = gtcrn::param_table();
    return (i < 0 || i >= (long)t.size()) ? nullptr : t[i].name.c_str();
}
long gtcrn_param_numel(long i) {
    const auto& t = gtcrn::param_table();
    return (i < 0 || i >= (long)t.size()) ? -1 : t[i].numel;
}
long gtcrn_param_offset(long i) {
    const auto& t = gtcrn::param_table();
    return (i < 0 || i >= (long)t.size()) ? -1 : t[i].offset;
}

// Host-only view of the packer for the CPU test-suite (no device needed): fills the slot-space
// buffers the kernels consume.  Sizes: gtcrn_pack_sizes().
void gtcrn_pack_sizes(long* n_floats, long* n_ints) {
    if (n_floats) *n_floats = gtl::P_FLOATS;
    if (n_ints) *n_ints = gtl::P_INTS;
}
int gtcrn_pack_params_host(const float* h_params, long n, float* h_f, int* h_i) {
    std::string err;
    if (gtcrn::pack_params(h_params, n, h_f, h_i, err) != 0) return fail(GTCRN_ERR_ARG, err);
    return 0;
}
int gtcrn_pack_params_quant_host(const float* h_params, long n, float* h_f, int* h_i) {
    int rc = gtcrn_pack_params_host(h_params, n, h_f, h_i);
    if (rc == 0) gtcrn::quantize_packed(h_f);
    return rc;
}
float gtcrn_round_to_half(float x) { return gtcrn::round_to_half(x); }

int gtcrn_model_create(gtcrn_model** out, const float* h_params, long n_floats, int device) {
    if (!out || !h_params) return fail(GTCRN_ERR_ARG, "null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(GTCRN_ERR_DEVICE, "no HIP device: this library has no CPU path");
    if (device < 0 || device >= ndev) return fail(GTCRN_ERR_ARG, "device ordinal out of range");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(GTCRN_ERR_DEVICE, std::string("built for gfx950 only, device is ") + prop.gcnArchName);
    HIP_TRY(hipSetDevice(device));
    gtcrn_model* m = new gtcrn_model();
    m->device = device;
    hipError_t e = hipMalloc(&m->d_pf, sizeof(float) * gtl::P_FLOATS);
    if (e == hipSuccess) e = hipMalloc(&m->d_pfq, sizeof(float) * gtl::P_FLOATS);
    if (e == hipSuccess) e = hipMalloc(&m->d_pi, sizeof(int) * gtl::P_INTS);
    if (e == hipSuccess) e = hipMalloc(&m->d_twid, sizeof(float) * 1024);
    if (e == hipSuccess) e = hipMalloc(&m->d_pref, sizeof(int) * 1025);
    if (e == hipSuccess) e = hipMalloc(&m->d_ptr8, sizeof(float*) * 8);
    if (e != hipSuccess) {
        gtcrn_model_destroy(m);
        return hip_fail(e, "hipMalloc(params)");
    }
    std::vector<float> tw(1024);
    for (int k = 0; k < 256; ++k) {
        tw[2 * k] = (float)std::cos(-2.0 * M_PI * k / 256.0);
        tw[2 * k + 1] = (float)std::sin(-2.0 * M_PI * k / 256.0);
        tw[512 + 2 * k] = (float)std::cos(-2.0 * M_PI * k / 512.0);
        tw[512 + 2 * k + 1] = (float)std::sin(-2.0 * M_PI * k / 512.0);
    }
    e = hipMemcpy(m->d_twid, tw.data(), sizeof(float) * 1024, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        gtcrn_model_destroy(m);
        return hip_fail(e, "hipMemcpy(twiddles)");
    }
    int rc = upload_params(m, h_params, n_floats);
    if (rc == 0) {
        int le = gtk::configure_kernels();
        if (le != 0) rc = hip_fail(static_cast<hipError_t>(le), "hipFuncSetAttribute(dynamic LDS)");
    }
    if (rc != 0) {
        gtcrn_model_destroy(m);
        return rc;
    }
    *out = m;
    return 0;
}

int gtcrn_model_set_params(gtcrn_model* m, const float* h_params, long n_floats) {
    int rc = check_model(m);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return upload_params(m, h_params, n_floats);
}

void gtcrn_model_destroy(gtcrn_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    (void)hipDeviceSynchronize();
    free_workspace(m);
    if (m->d_dbg) (void)hipFree(m->d_dbg);
    if (m->d_stamps) (void)hipFree(m->d_stamps);
    if (m->d_pf) (void)hipFree(m->d_pf);
    if (m->d_pfq) (void)hipFree(m->d_pfq);
    if (m->d_pi) (void)hipFree(m->d_pi);
    if (m->d_twid) (void)hipFree(m->d_twid);
    if (m->d_pref) (void)hipFree(m->d_pref);
    if (m->d_ptr8) (void)hipFree(m->d_ptr8);
    for (auto* v : {&m->timings, &m->ev_pool})
        for (auto& t : *v) {
            if (t.a) (void)hipEventDestroy(t.a);
            if (t.b) (void)hipEventDestroy(t.b);
        }
    delete m;
}

int gtcrn_model_reserve(gtcrn_model* m, int B, int T) {
    int rc = check_model(m);
    if (rc) return rc;
    if (B < 1 || T < 1) return fail(GTCRN_ERR_ARG, "B and T must be >= 1");
    return ensure_workspace(m, B, T, nullptr);
}

int gtcrn_make_window(int kind, float* h_w512) {
    if (!h_w512 || (kind != 0 && kind != 1)) return fail(GTCRN_ERR_ARG, "window kind must be 0 (sqrt-Hann) or 1 (Hann)");
    gtcrn::make_window(kind, h_w512);
    return 0;
}

long gtcrn_num_frames(long L) { return 1 + L / 256; }

// twiddles are per model; the standalone STFT entry points keep one immutable table per device, created once
// under a lock (the only process-wide state of the library; never modified after its creation)
namespace {
float* g_twid[16] = {nullptr};
std::mutex g_twid_mu;
int device_twiddles(float** out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) return fail(GTCRN_ERR_DEVICE, "device ordinal >= 16");
    std::lock_guard<std::mutex> lock(g_twid_mu);
    if (!g_twid[dev]) {
        std::vector<float> tw(1024);
        for (int k = 0; k < 256; ++k) {
            tw[2 * k] = (float)std::cos(-2.0 * M_PI * k / 256.0);
            tw[2 * k + 1] = (float)std::sin(-2.0 * M_PI * k / 256.0);
            tw[512 + 2 * k] = (float)std::cos(-2.0 * M_PI * k / 512.0);
            tw[512 + 2 * k + 1] = (float)std::sin(-2.0 * M_PI * k / 512.0);
        }
        float* d = nullptr;
        HIP_TRY(hipMalloc(&d, sizeof(float) * 1024));
        HIP_TRY(hipMemcpy(d, tw.data(), sizeof(float) * 1024, hipMemcpyHostToDevice));
        g_twid[dev] = d;
    }
    *out = g_twid[dev];
    return 0;
}
}  // namespace

extern "C" int gtcrn_device_twiddles_(float** out) { return device_twiddles(out); }   // for train.cpp

static int check_spec_layout(const void* p, long sb, long sf, long st) {
    // the kernels move the (re, im) pair with 8-byte accesses
    if ((reinterpret_cast<uintptr_t>(p) & 7) || (sb & 1) || (sf & 1) || (st & 1))
        return fail(GTCRN_ERR_ARG, "spectrogram base must be 8-byte aligned and its strides even (re/im pairs)");
    // inside one 16-frame chunk the kernels address with 32-bit element offsets (256 bins, 15 frames)
    const long span = 256 * (sf < 0 ? -sf : sf) + 15 * (st < 0 ? -st : st);
    if (span >= (1L << 31)) return fail(GTCRN_ERR_ARG, "spectrogram strides too large (a chunk must span < 2^31 floats)");
    return 0;
}

int gtcrn_stft(const float* d_wave, int B, long L, const float* d_win, float* d_spec, long sb, long sf, long st,
               void* stream) {
    if (!d_wave || !d_win || !d_spec || B < 1) return fail(GTCRN_ERR_ARG, "null pointer or B < 1");
    if (int rc0 = check_spec_layout(d_spec, sb, sf, st)) return rc0;
    if (L < 257) return fail(GTCRN_ERR_ARG, "reflect padding needs L > 256 samples");
    float* tw = nullptr;
    int rc = device_twiddles(&tw);
    if (rc) return rc;
    LAUNCH_TRY(gtk::launch_stft(d_wave, B, L, (int)gtcrn_num_frames(L), nullptr, d_win, tw, d_spec, sb, sf, st, nullptr,
                                (hipStream_t)stream));
    return 0;
}

int gtcrn_stft_frames(const float* d_wave, int B, long L, const float* d_win, float* d_frames, void* stream) {
    if (!d_wave || !d_win || !d_frames || B < 1) return fail(GTCRN_ERR_ARG, "null pointer or B < 1");
    if (L < 257) return fail(GTCRN_ERR_ARG, "reflect padding needs L > 256 samples");
    float* tw = nullptr;
    int rc = device_twiddles(&tw);
    if (rc) return rc;
    LAUNCH_TRY(gtk::launch_stft(d_wave, B, L, (int)gtcrn_num_frames(L), nullptr, d_win, tw, nullptr, 0, 0, 0, d_frames,
                                (hipStream_t)stream));
    return 0;
}

int gtcrn_istft(const float* d_spec, long sb, long sf, long st, int B, int T, const float* d_win, float* d_wave,
                void* stream) {
    if (!d_spec || !d_win || !d_wave || B < 1) return fail(GTCRN_ERR_ARG, "null pointer or B < 1");
    if (T < 2) return fail(GTCRN_ERR_ARG, "iSTFT needs T >= 2 frames");
    if (int rc0 = check_spec_layout(d_spec, sb, sf, st)) return rc0;
    float* tw = nullptr;
    int rc = device_twiddles(&tw);
    if (rc) return rc;
    LAUNCH_TRY(gtk::launch_istft(d_spec, sb, sf, st, B, T, nullptr, d_win, tw, d_wave, (hipStream_t)stream));
    return 0;
}

int gtcrn_forward_spec(gtcrn_model* m, const float* d_spec_in, long isb, long isf, long ist, float* d_spec_out,
                       long osb, long osf, long ost, int B, int T, void* stream) {
    int rc = check_model(m);
    if (rc) return rc;
    if (!d_spec_in || !d_spec_out) return fail(GTCRN_ERR_ARG, "null spectrogram pointer");
    if (B < 1 || T < 1) return fail(GTCRN_ERR_ARG, "B and T must be >= 1");
    if (int rc0 = check_spec_layout(d_spec_in, isb, isf, ist)) return rc0;
    if (int rc0 = check_spec_layout(d_spec_out, osb, osf, ost)) return rc0;
    hipStream_t s = (hipStream_t)stream;
    rc = ensure_workspace(m, B, T, s);
    if (rc) return rc;
    return run_model(m, d_spec_in, isb, isf, ist, d_spec_out, osb, osf, ost, B, T, nullptr, s);
}

static int forward_wave_impl(gtcrn_model* m, const float* d_wave, float* d_wave_out, int B, long L,
                             const int* d_lengths, const float* d_win, void* stream,
                             const gtk::Quant* q = nullptr) {
    int rc = check_model(m);
    if (rc) return rc;
    if (!d_wave || !d_wave_out || !d_win) return fail(GTCRN_ERR_ARG, "null pointer");
    if (B < 1) return fail(GTCRN_ERR_ARG, "B must be >= 1");
    if (L < 257) return fail(GTCRN_ERR_ARG, "reflect padding needs L > 256 samples");
    hipStream_t s = (hipStream_t)stream;
    const int T = (int)gtcrn_num_frames(L);
    rc = ensure_workspace(m, B, T, s);
    if (rc) return rc;
    // internal spectrograms are frame-major (B,T,257,2): sb = T*514, sf = 2, st = 514
    const long sb = (long)T * 514, sf = 2, st = 514;
    Timer tm(m, s);
    tm.begin(K_FRONT);
    LAUNCH_TRY(gtk::launch_front(d_wave, L, nullptr, 0, 0, 0, B, T, d_lengths, d_win, m->d_twid, q ? m->d_pfq : m->d_pf,
                                 m->d_pi, m->d_spec_a, m->d_en0, m->d_en[0], s, q));
    tm.end();
    rc = run_model(m, m->d_spec_a, sb, sf, st, m->d_spec_b, sb, sf, st, B, T, nullptr, s, d_lengths, q, true);
    if (rc) return rc;
    tm.begin(K_ISTFT);
    LAUNCH_TRY(gtk::launch_istft(m->d_spec_b, sb, sf, st, B, T, d_lengths, d_win, m->d_twid, d_wave_out, s));
    tm.end();
    return 0;
}

int gtcrn_forward_wave(gtcrn_model* m, const float* d_wave, float* d_wave_out, int B, long L, const float* d_win,
                       void* stream) {
    return forward_wave_impl(m, d_wave, d_wave_out, B, L, nullptr, d_win, stream);
}

int gtcrn_forward_wave_var(gtcrn_model* m, const float* d_wave, float* d_wave_out, int B, long Lmax,
                           const int* d_lengths, const float* d_win, void* stream) {
    if (!d_lengths) return fail(GTCRN_ERR_ARG, "gtcrn_forward_wave_var: null lengths");
    return forward_wave_impl(m, d_wave, d_wave_out, B, Lmax, d_lengths, d_win, stream);
}

static int quant_opts(float in_scale, float out_scale, gtk::Quant* q) {
    if (in_scale < 0.f || out_scale < 0.f) return fail(GTCRN_ERR_ARG, "quantiser scales must be >= 0 (0 = fp16 boundary)");
    q->in_step = in_scale / 255.0f;     // calibration maps [-scale/2, scale/2] onto the 255 int8 steps
    q->out_step = out_scale / 255.0f;
    return 0;
}

int gtcrn_forward_spec_quant(gtcrn_model* m, const float* d_spec_in, long isb, long isf, long ist, float* d_spec_out,
                             long osb, long osf, long ost, int B, int T, float in_scale, float out_scale, void* stream) {
    int rc = check_model(m);
    if (rc) return rc;
    if (!d_spec_in || !d_spec_out) return fail(GTCRN_ERR_ARG, "null spectrogram pointer");
    if (B < 1 || T < 1) return fail(GTCRN_ERR_ARG, "B and T must be >= 1");
    if (int rc0 = check_spec_layout(d_spec_in, isb, isf, ist)) return rc0;
    if (int rc0 = check_spec_layout(d_spec_out, osb, osf, ost)) return rc0;
    gtk::Quant q;
    if ((rc = quant_opts(in_scale, out_scale, &q))) return rc;
    hipStream_t s = (hipStream_t)stream;
    rc = ensure_workspace(m, B, T, s);
    if (rc) return rc;
    return run_model(m, d_spec_in, isb, isf, ist, d_spec_out, osb, osf, ost, B, T, nullptr, s, nullptr, &q);
}

int gtcrn_forward_wave_quant(gtcrn_model* m, const float* d_wave, float* d_wave_out, int B, long L, const float* d_win,
                             float in_scale, float out_scale, void* stream) {
    gtk::Quant q;
    if (int rc = quant_opts(in_scale, out_scale, &q)) return rc;
    return forward_wave_impl(m, d_wave, d_wave_out, B, L, nullptr, d_win, stream, &q);
}

size_t gtcrn_stream_state_bytes(void) { return sizeof(float) * gtk::ST_FLOATS; }

int gtcrn_stream_reset(gtcrn_model* m, void* d_state, int nstreams, void* stream) {
    int rc = check_model(m);
    if (rc) return rc;
    if (!d_state || nstreams < 1) return fail(GTCRN_ERR_ARG, "null state or nstreams < 1");
    HIP_TRY(hipMemsetAsync(d_state, 0, gtcrn_stream_state_bytes() * nstreams, (hipStream_t)stream));
    return 0;
}

int gtcrn_stream_step(gtcrn_model* m, void* d_state, const float* d_spec_t, long isb, long isf, long ist,
                      float* d_spec_out_t, long osb, long osf, long ost, int nstreams, int nframes, void* stream) {
    int rc = check_model(m);
    if (rc) return rc;
    if (!d_state || !d_spec_t || !d_spec_out_t) return fail(GTCRN_ERR_ARG, "null pointer");
    if (nstreams < 1 || nframes < 1) return fail(GTCRN_ERR_ARG, "nstreams and nframes must be >= 1");
    if (int rc0 = check_spec_layout(d_spec_t, isb, isf, ist)) return rc0;
    if (int rc0 = check_spec_layout(d_spec_out_t, osb, osf, ost)) return rc0;
    hipStream_t s = (hipStream_t)stream;
    rc = ensure_workspace(m, nstreams, nframes, s);
    if (rc) return rc;
    return run_model(m, d_spec_t, isb, isf, ist, d_spec_out_t, osb, osf, ost, nstreams, nframes, (float*)d_state, s);
}

static int state_convert(gtcrn_model* m, void* d_state, int nstreams, float* conv, float* tra,
                         float* const* tcn8, int dir, void* stream) {
    int rc = check_model(m);
    if (rc) return rc;
    if (!d_state || !conv || !tra || !tcn8 || nstreams < 1) return fail(GTCRN_ERR_ARG, "null pointer or nstreams < 1");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(m->d_ptr8, tcn8, sizeof(float*) * 8, hipMemcpyHostToDevice, s));
    LAUNCH_TRY(gtk::launch_state_convert((float*)d_state, nstreams, conv, tra, m->d_ptr8, m->d_pi, dir, s));
    return 0;
}

int gtcrn_stream_import(gtcrn_model* m, void* d_state, int nstreams, const float* d_conv_cache,
                        const float* d_tra_cache, const float* const* d_tcn_cache8, void* stream) {
    return state_convert(m, d_state, nstreams, const_cast<float*>(d_conv_cache), const_cast<float*>(d_tra_cache),
                         const_cast<float* const*>(reinterpret_cast<const float* const*>(d_tcn_cache8)), 0, stream);
}

int gtcrn_stream_export(gtcrn_model* m, const void* d_state, int nstreams, float* d_conv_cache, float* d_tra_cache,
                        float* const* d_tcn_cache8, void* stream) {
    return state_convert(m, const_cast<void*>(d_state), nstreams, d_conv_cache, d_tra_cache, d_tcn_cache8, 1, stream);
}

int gtcrn_stream_conv2d(const float* d_x, const float* d_cache, const float* d_w, const float* d_bias, float* d_y,
                        float* d_cache_out, int B, int Cin, int Cout, int T, int F, int kt, int kf, int dt, int df,
                        int pad_f, int groups, int transposed, void* stream) {
    if (!d_x || !d_w || !d_y) return fail(GTCRN_ERR_ARG, "null pointer");
    // pad_f may be negative: the reference pads by (kF-1)*dF - F_pad, which crops when kF == 1 (convolution.py:243-250)
    if (B < 1 || Cin < 1 || Cout < 1 || T < 1 || F < 1 || kt < 1 || kf < 1 || dt < 1 || df < 1)
        return fail(GTCRN_ERR_ARG, "bad shape");
    const int H = (kt - 1) * dt;
    if (H > 0 && (!d_cache || !d_cache_out)) return fail(GTCRN_ERR_ARG, "a causal kernel taller than 1 needs a cache");
    if (transposed && groups != 1) return fail(GTCRN_ERR_ARG, "transposed form supports groups == 1 only");
    if (!transposed && (groups < 1 || Cin % groups || Cout % groups)) return fail(GTCRN_ERR_ARG, "bad groups");
    const int Fout = transposed ? F - 2 * pad_f + df * (kf - 1) : F + 2 * pad_f - df * (kf - 1);
    if (Fout < 1) return fail(GTCRN_ERR_ARG, "empty output");
    LAUNCH_TRY(gtk::launch_conv2d_causal(d_x, d_cache, d_w, d_bias, d_y, d_cache_out, B, Cin, Cout, T, F, kt, kf, dt,
                                         df, pad_f, groups, transposed, Fout, (hipStream_t)stream));
    return Fout;
}

int gtcrn_debug_enable(gtcrn_model* m, int on) {
    int rc = check_model(m);
    if (rc) return rc;
    m->debug = on != 0;
    m->debug_keep_fused = on == 2;
    return 0;
}

int gtcrn_var_spans_enable(gtcrn_model* m, int on) {
    int rc = check_model(m);
    if (rc) return rc;
    m->var_spans = on != 0;
    return 0;
}

int gtcrn_stream_streams_per_workgroup(int nstreams) {
    if (nstreams < 1) return fail(GTCRN_ERR_ARG, "gtcrn_stream_streams_per_workgroup: nstreams >= 1");
    return stream_wide_pays(nstreams) ? gtk::stream_wide_streams() : 4;
}

int gtcrn_stream_form(gtcrn_model* m, int form) {
    int rc = check_model(m);
    if (rc) return rc;
    if (form < 0 || form > 3)
        return fail(GTCRN_ERR_ARG, "gtcrn_stream_form: 0 (one launch, form by stream count), 1 (three launches), 2 / 3 (one launch: "
                                   "four / seven streams per workgroup)");
    m->stream_form = form;
    return 0;
}

long gtcrn_debug_tap(gtcrn_model* m, const char* name, int b, float* h_dst, long cap) {
    int rc = check_model(m);
    if (rc) return rc;
    if (!name || !h_dst) return fail(GTCRN_ERR_ARG, "null argument");
    const int B = m->last_B, T = m->last_T;
    if (B < 1 || b < 0 || b >= B) return fail(GTCRN_ERR_ARG, "no forward recorded or batch index out of range");
    if (m->last_quant)
        return fail(GTCRN_ERR_STATE, "the last forward was the int8/fp16 variant: its hand-off tensors are fp16 records, "
                                     "the stage taps read fp32");
    if (m->last_fused_stream)
        return fail(GTCRN_ERR_STATE, "the last forward was a single-launch streaming step, which keeps no hand-off "
                                     "tensors: enable debug before the step to run the three-launch form");
    const std::string nm(name);
    const long bt = (long)B * T;
    const float* src = nullptr;
    int F = 33, perm = -1;
    if (nm == "en0") { src = m->d_en0; F = 65; perm = 0; }
    else if (nm == "en1") { src = m->d_en[0]; perm = 1; }
    else if (nm == "en2") { src = m->d_en[1]; perm = 2; }
    else if (nm == "en3") { src = m->d_en[2]; perm = 3; }
    else if (nm == "en4") { src = m->d_en[3]; perm = 4; }
    else if (nm == "gtcn1") { src = m->d_g1; perm = 4; }
    else if (nm == "gtcn2") { src = m->d_g2; perm = 4; }
    else if (nm == "de0" || nm == "de1" || nm == "de2" || nm == "de3" || nm == "de4") {
        if (!m->debug || !m->d_dbg || m->dbg_cap_bt < bt)
            return fail(GTCRN_ERR_STATE, "decoder taps need gtcrn_debug_enable(m,1) before the forward");
        const int j = nm[2] - '0';
        if (j < 3) { src = m->d_dbg + (long)j * bt * 528; perm = 5 + j; }
        else if (j == 3) { src = m->d_dbg + 3 * bt * 528; F = 65; perm = 8; }
        else {
            // de4 is stored as (B,2,T,129) already
            const long nel = 2L * T * 129;
            if (cap < nel) return fail(GTCRN_ERR_ARG, "destination too small");
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipMemcpy(h_dst, m->d_dbg + 3 * bt * 528 + bt * 65 * 16 + (long)b * nel, sizeof(float) * nel,
                              hipMemcpyDeviceToHost));
            return nel;
        }
    } else {
        return fail(GTCRN_ERR_ARG, "unknown tap name");
    }
    const long nel = 16L * T * F;
    if (cap < nel) return fail(GTCRN_ERR_ARG, "destination too small");
    std::vector<float> tmp(nel);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(tmp.data(), src + (long)b * T * F * 16, sizeof(float) * nel, hipMemcpyDeviceToHost));
    if (nm == "gtcn2") {  // stored as gtcn2(x) + en4: recover the stage output for the tap
        std::vector<float> e4(nel);
        HIP_TRY(hipMemcpy(e4.data(), m->d_en[3] + (long)b * T * F * 16, sizeof(float) * nel, hipMemcpyDeviceToHost));
        for (long i = 0; i < nel; ++i) tmp[i] -= e4[i];
    }
    const int* pm = m->h_pi.data() + gtl::I_PERM + perm * 16;
    for (int t = 0; t < T; ++t)
        for (int f = 0; f < F; ++f)
            for (int s = 0; s < 16; ++s) h_dst[((long)pm[s] * T + t) * F + f] = tmp[((long)t * F + f) * 16 + s];
    return nel;
}

long gtcrn_debug_stamps(gtcrn_model* m, int kernel, unsigned long long* h_dst, long cap) {
    int rc = check_model(m);
    if (rc) return rc;
    if (kernel < 0 || kernel > 3 || !h_dst) return fail(GTCRN_ERR_ARG, "kernel index 0..3 (encoder, gtcn1, gtcn2, decoder)");
    if (!m->d_stamps) return fail(GTCRN_ERR_STATE, "enable debug before the forward");
    const long n = (long)m->last_B * 16;
    if (cap < n) return fail(GTCRN_ERR_ARG, "destination too small");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h_dst, m->d_stamps + (long)kernel * m->stamps_cap_b * 16, sizeof(unsigned long long) * n,
                      hipMemcpyDeviceToHost));
    return n;
}

namespace {
// the self-tests own their scratch buffers and the device selection: both are put back on EVERY exit path
struct DevBuf {
    float* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
};
struct DeviceScope {
    int prev = -1;
    DeviceScope() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};
}  // namespace

static int pcm16_convert(int device, const void* src, void* dst, long n, int dir, void* stream, const char* who) {
    if (!dst || !src) return fail(GTCRN_ERR_ARG, std::string(who) + ": null pointer");
    if (n <= 0 || (n & 7) || ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15))
        return fail(GTCRN_ERR_ARG, std::string(who) + ": 16-byte aligned device pointers and a sample count that is a multiple of 8");
    DeviceScope scope;              // (a C caller's current device is put back)
    HIP_TRY(hipSetDevice(device));
    LAUNCH_TRY(gtk::launch_pcm16_convert(src, dst, n, dir, static_cast<hipStream_t>(stream)));
    return 0;
}
int gtcrn_pcm16_to_f32(int device, const short* d_pcm, float* d_wave, long n, void* stream) {
    return pcm16_convert(device, d_pcm, d_wave, n, 0, stream, "gtcrn_pcm16_to_f32");
}
int gtcrn_f32_to_pcm16(int device, const float* d_wave, short* d_pcm, long n, void* stream) {
    return pcm16_convert(device, d_wave, d_pcm, n, 1, stream, "gtcrn_f32_to_pcm16");
}

int gtcrn_selftest_mfma(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(GTCRN_ERR_DEVICE, "no HIP device");
    DeviceScope restore;
    HIP_TRY(hipSetDevice(device));
    float hA[64], hB[64], hC[256], hD[256], ref[256];
    for (int i = 0; i < 16; ++i)
        for (int k = 0; k < 4; ++k) hA[i * 4 + k] = (float)(1 + i * 5 + k * 3);       // asymmetric integers
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 16; ++j) hB[k * 16 + j] = (float)(2 + k * 7 - j * 2 + (j * j) % 5);
    for (int i = 0; i < 256; ++i) hC[i] = (float)((i * 37) % 101 - 50);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float s = hC[i * 16 + j];
            for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j];
            ref[i * 16 + j] = s;
        }
    DevBuf bA, bB, bC, bD;
    HIP_TRY(bA.alloc(sizeof(hA))); HIP_TRY(bB.alloc(sizeof(hB)));
    HIP_TRY(bC.alloc(sizeof(hC))); HIP_TRY(bD.alloc(sizeof(hD)));
    float *dA = bA.p, *dB = bB.p, *dC = bC.p, *dD = bD.p;
    HIP_TRY(hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dC, hC, sizeof(hC), hipMemcpyHostToDevice));
    LAUNCH_TRY(gtk::launch_selftest(dA, dB, dC, dD, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
    if (bad) return fail(GTCRN_ERR_DEVICE, "MFMA 16x16x4 f32 lane map differs from the assumed one (" +
                                               std::to_string(bad) + " of 256 elements)");
    return 0;
}

int gtcrn_selftest_split3(int device, const float* h_x, long n, float* h_planes, float* h_joined, const float* h_A,
                          const float* h_B, float* h_D) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(GTCRN_ERR_DEVICE, "no HIP device");
    if (!h_x || !h_planes || !h_joined || n < 4 || (n & 3)) return fail(GTCRN_ERR_ARG, "n must be a positive multiple of 4");
    if ((h_A || h_B || h_D) && !(h_A && h_B && h_D)) return fail(GTCRN_ERR_ARG, "A, B and D come together");
    DeviceScope restore;
    HIP_TRY(hipSetDevice(device));
    DevBuf bx, bp, bj, bm;             // (freed on every exit path, ADVICE r4: a failing call leaked all four)
    HIP_TRY(bx.alloc(sizeof(float) * n));
    HIP_TRY(bp.alloc(sizeof(float) * 3 * n));
    HIP_TRY(bj.alloc(sizeof(float) * n));
    HIP_TRY(bm.alloc(sizeof(float) * (512 + 512 + 256)));
    float *dx = bx.p, *dp = bp.p, *dj = bj.p, *dm = bm.p;
    HIP_TRY(hipMemcpy(dx, h_x, sizeof(float) * n, hipMemcpyHostToDevice));
    if (h_A) {
        HIP_TRY(hipMemcpy(dm, h_A, sizeof(float) * 512, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dm + 512, h_B, sizeof(float) * 512, hipMemcpyHostToDevice));
    }
    LAUNCH_TRY(gtk::launch_selftest_split3(dx, n, dp, dj, h_A ? dm : nullptr, dm + 512, dm + 1024, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h_planes, dp, sizeof(float) * 3 * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h_joined, dj, sizeof(float) * n, hipMemcpyDeviceToHost));
    if (h_D) HIP_TRY(hipMemcpy(h_D, dm + 1024, sizeof(float) * 256, hipMemcpyDeviceToHost));
    return 0;
}

int gtcrn_timing_enable(gtcrn_model* m, int on) {
    int rc = check_model(m);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    for (auto& t : m->timings) m->ev_pool.push_back(t);
    m->timings.clear();
    m->timing = on != 0;
    m->timing_only = on >= 2 ? on - 2 : -1;
    if (m->timing_only >= kNumKernels) return fail(GTCRN_ERR_ARG, "gtcrn_timing_enable: no such kernel index");
    return 0;
}

int gtcrn_timing_kernels(void) { return kNumKernels; }

int gtcrn_timing_read(gtcrn_model* m, int idx, char* name, int name_cap, float* ms, int* launches) {
    if (idx < 0) {   // name query only (no model needed): idx = -1 - k
        const int k = -1 - idx;
        if (k >= kNumKernels || !name || name_cap < 1) return fail(GTCRN_ERR_ARG, "timing index out of range");
        std::strncpy(name, kKernelNames[k], name_cap - 1);
        name[name_cap - 1] = 0;
        if (ms) *ms = 0.f;
        if (launches) *launches = 0;
        return 0;
    }
    int rc = check_model(m);
    if (rc) return rc;
    if (idx < 0 || idx >= kNumKernels || !ms) return fail(GTCRN_ERR_ARG, "timing index out of range");
    double sum = 0.0;
    int n = 0;
    for (auto& t : m->timings)
        if (t.kernel == idx) {
            float v = 0.f;
            HIP_TRY(hipEventSynchronize(t.b));
            HIP_TRY(hipEventElapsedTime(&v, t.a, t.b));
            sum += v;
            ++n;
        }
    *ms = n ? (float)(sum / n) : 0.f;
    if (launches) *launches = n;
    if (name && name_cap > 0) {
        std::strncpy(name, kKernelNames[idx], name_cap - 1);
        name[name_cap - 1] = 0;
    }
    return 0;
}

}  // extern "C"
