// qk_engine.hip -- host side of the receive-DSP engine: plans the reference's decimation / demodulation
// chain (quisk_process_decimate, quisk_process_demodulate, quisk_process_samples in quisk.c) for N
// receivers, owns the streams in HBM and launches the kernels of qk_kernels.h.  Exposes the C ABI of
// include/qk_engine.h.  No CPU fallback: every entry point that computes needs a HIP device.
#include "../../include/qk_engine.h"
#include <hip/hip_ext.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include "qk_kernels.h"
#include "qk_fft.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

extern "C" void qk_phase_step(int tune_hz, int sample_rate, double *re, double *im);
extern "C" void qk_phase_tables(double pre, double pim, double *C, double *B, double *A, int nA);

#define QK_COEF_TABLE(name, n) static const double name[n]
#include "qk_coefs.inc"

// librccl is loaded on first use: a one-GPU user of the library does not need it present.
namespace {
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
} g_rccl;
std::mutex g_rccl_mu;
}
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
// Flags of the events that order one stream behind another
static unsigned sync_event_flags() { return hipEventDisableTiming; }
#define HIPCHECK(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess)                                                                            \
            return fail(QK_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));                     \
    } while (0)

namespace {

constexpr int DEFAULT_TAP_CAP = 1024;
constexpr int FFT_BATCH = 8;          // display-spectrum frames per four-step launch
constexpr int FFT_GROUPS = 2;         // frame groups of the four-step row pass, each with its own partial average
constexpr int EV_POOL = 2048;
constexpr int QK_MAX_ROUTES = 4;       // stereo routes (split RX/TX, multirx play) per engine: two AGC slots each behind the receivers'
constexpr int AGC_SKIP_CLASS = 7;      // an AGC class index that never has samples: a receiver whose AGC runs in a route's slot

enum StageKind { ST_FIR, ST_INTERP, ST_HBINT, ST_RESAMP, ST_RXF, ST_FMDISC, ST_IIR, ST_FRAC };
enum ChainClass { CL_SSB = 0, CL_CW, CL_AM, CL_FM, CL_DGTW, CL_DGTIQ, CL_COUNT };    // CL_DGTW: DGT-U/L filtered at the decimated rate;
                                                                                     // CL_DGTIQ: complex (stereo) audio

// The reference's filter objects that more than one mode uses (Storage[bank] of quisk_process_demodulate, quisk.c:1730-1747;
// cRxFilterOut / dRxFilterOut rings, quisk.c:1124-1127, 1161-1165; the statics of quisk_process_samples, quisk.c:2178-2180;
// cFracDecim's statics, quisk.c:575-576), or whose decimation counter a receiver keeps while it is in another mode.  A stage
// that stands for one of them carries its id, so that a receiver changing chain class takes the object's contents along.
enum ObjId { OB_NONE = -1, OB_HB5 = 0, OB_HB4, OB_D48, OB_CRX, OB_DRX, OB_HB6, OB_HB7, OB_A24P3, OB_FRAC, OB_PHB0, OB_PHB1, OB_PHB2, OB_COUNT };

struct Stream {
    double *d = nullptr;
    int planes = 1, hist = 0, cap = 0, n = 0;
    bool f32 = false;          // the elements are floats (same element strides on the same allocation): the fp32 mode's front-chain
                               // streams between the packed front end and the float FIRs
    long long rx_stride() const { return (long long)planes * (hist + cap); }
    StreamView view() const { return StreamView{d, rx_stride(), hist + cap, hist}; }
};

struct Stage {
    StageKind kind;
    int in, out;               // stream indices within the owning chain (in == -1: the chain's input stream)
    const double *h = nullptr; // device taps (ST_FIR: permuted [D][TQP], see fir_tile)
    const float *hf = nullptr; // ST_FIR: the same taps in float (fp32 mode)
    int T = 0, D = 1, L = 1, TQP = 0;
    int phase = 0;             // decimation counter / resampler index carried between blocks
    int rxf_kind = 0, iir_kind = 0;
    double a0 = 0, a1 = 0, b1 = 0, fs = 0;
    double *iir_state = nullptr;
    int timer = -1;            // QK_KT_* bucket
    int hook = 0;              // audio hooks behind this stage: 1 = auto-notch and SSB squelch, 2 = auto-notch only (quisk.c:1792, 1829-1834, 1909)
    int hook_div = 1;          // the hooks' sample rate is decim_srate / hook_div (quisk_filter_srate; 48 k behind the FM chain)
    bool delayed = false;      // this stage reads its input SSQ_N samples late while the SSB squelch is on (d_delay, quisk.c:1833)
    // Two stages on ONE state object of the reference (filtDecim5S, quisk.c:1701 and 1713): both read the same stream, the
    // second user's samples of a block stored behind the first user's, and both move the same decimation counter
    bool in_after = false, out_after = false;   // this stage's input / output starts behind the first user's samples of the block
    int phase_share = -1;      // index of the stage whose `phase` is this stage's counter too
    int measure_div = 0;       // > 0: receiver 0's output of this stage is what measure_audio sums (quisk.c:1787-1790 ...); the stage's
                               // rate is decim_srate / measure_div (quisk_filter_srate)
    long long total_in = 0;    // ST_RXF: samples this stage has taken in since the engine was made (ring positions, see RingImage)
    int obj = OB_NONE;         // the reference's filter object this stage stands for, if a receiver can carry it between classes
};

struct Chain {
    bool built = false;
    std::vector<Stream> streams;
    std::vector<Stage> stages;
    std::vector<int> members;  // receivers in this class
    int *d_members = nullptr;
    int aud = -1;              // stream index of the 48 kHz-class mono audio
    int play = -1;             // stream index of the audio at the playback rate (AGC input)
    int n_out = 0;             // audio samples of the last block
    int shared = -1, shared_n1 = 0;   // front chain: the stream two stages share as one ring (see Stage::in_after), the first user's count
    double *front_alt = nullptr; // front chain only: ping-pong twin of streams[0] (the front end writes block b+1 while the chain reads block b)
    int front_last_n = 0;       // samples in the twin (the last non-empty block)
    double *play_alt = nullptr; // ping-pong twin of streams[play]: the AGC of block b runs beside the chain of block b+1
    // fractional decimator (cFracDecim, quisk.c:570-613): the position recurrence runs on the host
    double frac_dindex = 1.0;
    int *d_frac_idx = nullptr; double *d_frac_pos = nullptr; int frac_cap = 0;
    // the positions of a block go to the device through a small ring of pinned staging buffers
#define QK_FRAC_SLOTS 8
    struct FracSlot { int *idx = nullptr; double *pos = nullptr; hipEvent_t ev = nullptr; bool used = false; };
    FracSlot frac_slot[QK_FRAC_SLOTS];
    int frac_next = 0;
};

struct DevTables { std::vector<double> host; double *dev = nullptr; };
struct FirTab { const double *hp = nullptr; const float *hpf = nullptr; int T = 0, D = 1, TQP = 0; };     // decimating FIR: permuted taps on the device
constexpr int FT_NTH = 128, FT_R = 5;      // k_fir_tiled / k_rxf_tiled: lanes per workgroup, outputs per lane

}  // namespace

struct FftPlan;
struct Bluestein;
struct AnyFft;
struct qk_engine {
    qk_config cfg{};
    std::mutex mu;
    hipStream_t stream = nullptr;
    // Three-stage pipeline over blocks, one HIP stream per stage:
    //   fe_stream  : input copy/unpack, display spectrum, front end (mixer + half-bands), mixer state
    //   stream     : FIR chain, demodulators, audio chain of every class
    //   agc_stream : the serial AGC pass (one wave per receiver)
    // so the front end of block b+1 runs beside the chain of block b and the AGC of block b-1.
    hipStream_t fe_stream = nullptr;
    hipStream_t agc_stream = nullptr;
    //   apply_stream : k_agc_apply (gain x delayed sample, limiter) of block b beside the gain pass of block b+1: the gain
    //                  pass is the pipeline's critical loop, and the 17 us of the apply pass do not belong in it
    hipStream_t apply_stream = nullptr; bool last_apply_own = false; hipEvent_t ev_gain = nullptr; double *d_gains2 = nullptr; int *d_agc_started2 = nullptr;
    hipEvent_t ev_chain = nullptr, ev_agc[2] = {nullptr, nullptr};
    hipEvent_t ev_front = nullptr, ev_lane[2] = {nullptr, nullptr};      // fork / join of the second lane of class chains
    hipEvent_t ev_fe = nullptr, ev_fe2 = nullptr;      // front end (and head copy) of this block done (alternating)
    bool in_on_device = false;
    double2 *d_in_alt = nullptr; hipEvent_t ev_in[2] = {nullptr, nullptr}; bool in_used[2] = {false, false}; int in_slot = 0;   // short blocks: two
                                                            // pinned input blocks, so that a call does not wait for the previous block's last readers
    hipEvent_t ev_audio = nullptr; bool audio_ev = false;   // short blocks: recorded behind the kernels that write the audio rows (qk_pull_audio* wait for it,
                                                            // not for the tail copies and the spectrum's frames enqueued behind it)
    hipEvent_t ev_read[2] = {nullptr, nullptr};        // the chains have finished reading front-stream twin q
    bool read_pending[2] = {false, false};
    int fe_blocks = 0;                                 // blocks in which the front end produced output (twin parity)
    // QK_TIMELINE=<file>: eight stream marks per block over a window of blocks, written out as csv when the engine goes (a
    // measuring aid: where a block's stages sit in time against each other; profiles/tools/stream_marks.py reads it)
    std::vector<hipEvent_t> tl_ev; int tl_first = 0, tl_n = 0; std::string tl_path;
    long long pipe_blocks = 0;                         // blocks run by the receivers' pipeline: selects every ping-pong set (events, gains,
                                                       // squelch flags).  Not stats.blocks, which qk_reset_stats zeroes between blocks.
    bool agc_pending = false;
    int nrx = 0, S = 0, tap_cap = DEFAULT_TAP_CAP;
    bool fp32 = false;         // qk_config.reserved[1]: the front end, the tiled FIRs and the display FFT compute in float
    float *d_coefs_f = nullptr, *d_taps_f = nullptr;
    int fe_rate = 0, decim_srate = 0;
    int last_m_max = 0;        // audio samples per receiver of the previous block (0: none yet)
    int fe_slots = 2304;       // resident one-wave workgroups of k_frontend_wave on the whole chip (occupancy query)
    int fe_share = 100;        // percent of the resident slots the front end may occupy (QK_FE_SHARE)
    int fe_rounds = 2;         // front-end workgroups launched per resident slot (QK_FE_ROUNDS)
    bool fuse_chains = false;  // short blocks (small_mode; QK_FUSE overrides): a class chain, with the front chain's FIRs in front of it where
                               // one class holds every receiver, as one launch (k_chain_fused)
    FusedArgs fz_front{};      // short blocks: the front chain's FIR stages wait here for the class chain's launch (fz_defer), so that the
    bool fz_defer = false;     // decimators behind the front end and the whole class chain are ONE launch
    int fuse_max_in = 4096;    // ... up to this many samples behind the decimators, all receivers of the chain together (QK_FUSE_MAX)
    bool small_mode = false;   // short blocks (max_block <= 4096, QK_SMALL_MODE): one HIP stream instead of four (a block is a string of
                               // dependent launches: every hop between streams is an event packet's latency and buys nothing) and
                               // the input block and the audio rows in pinned host memory that the kernels read and write
                               // themselves (no copy commands, no pageable-memory staging)
    bool fe_packed = true;     // fp32 mode: the packed-FP32 front end k_frontend_pk (QK_FE_PACKED=0: the plane-per-lane float form)
    int rate_case = 0;
    // input
    double2 *d_in = nullptr;
    unsigned char *d_raw = nullptr; size_t raw_cap = 0;
    // mixer
    int H = 0, nA = 0;
    double2 *d_mixhist[2] = {nullptr, nullptr}, *d_v0[2] = {nullptr, nullptr};
    double2 *d_tabA = nullptr, *d_tabB = nullptr, *d_tabC = nullptr;
    int *d_mix_on = nullptr;
    int pp = 0;                // ping-pong selector
    long long N0 = 0;
    std::vector<int> tune, mode, cls;
    // coefficient tables
    double *d_coefs = nullptr;
    const double *h_dec5 = nullptr, *h_rs65 = nullptr, *h_a24p4 = nullptr;      // raw tables (resampler, interpolator)
    FirTab f_dec2, f_dec3, f_dec5, f_hbdec, f_s53, f_s111, f_s133, f_s167, f_s185, f_a24p3, f_a24p6, f_fmhp;
    // front chain (shared by all receivers): stream 0 = front-end output
    Chain front;
    Chain cls_chain[CL_COUNT];
    // per receiver
    double *d_taps = nullptr; int *d_ntaps = nullptr, *d_mode = nullptr, *d_cls = nullptr; double *d_release_gain = nullptr;
    std::vector<int> ntaps, bandwidth, start_offset;
    // what a receiver carries between chain classes (migrate_receiver): per object the last `need` input samples of the
    // reference's filter object, [rx][2 planes][need], and its decimation counter; cFracDecim's position
    // The band-pass rings as the reference holds them (bufFilterI/Q and bufFilterC with their indexFilter, quisk.c:1124-1127,
    // 1161-1165), kept on the host and brought up to date only when a receiver's filter LENGTH changes or it changes class:
    // with a constant length the ring is a plain FIR history, which is how the kernels keep it; at a new length the ring's
    // cells are read in another order (and cells beyond the old length come back into play), see ring_apply
    struct RingImage { std::vector<double> cell; int idx = 0, N = 0; long long cnt = 0; bool active = false; };
    std::vector<RingImage> rings;      // [rx][2]: cRxFilterOut's, dRxFilterOut's
    double *d_obj[OB_COUNT] = {nullptr}; int obj_need[OB_COUNT] = {0};
    std::vector<int> obj_phase; std::vector<double> obj_frac;
    // FM squelch and kill_audio: d_sq_state [rx][3], d_sq_flag [2][rx] (one set per block in flight towards the AGC)
    double squelch_level = -999.0; int kill_audio = 0; double *d_sq_state = nullptr, *d_sq_partial = nullptr; int *d_sq_flag = nullptr, sq_pstride = 0;
    AgcState *d_agc = nullptr; double *d_gains = nullptr, *d_agc_mag = nullptr; int gain_stride = 0;   // d_agc_mag: [rx][gain_stride] cabs of complex play streams
    int *d_agc_started = nullptr; std::vector<char> agc_started;    // per receiver: has process_agc run once?
    int *h_agc_abort = nullptr, *d_agc_abort = nullptr;             // pinned word k_agc_gain sets when a hand-over timed out
    double agc_release_time = 1.0;
    // input conditioning (DCremove, sound.c:165-230; spectrum inversion, quisk.c:2315-2320)
    int dc_bw = 0, invert = 0;
    double dc_alpha = 0.95; int dc_alpha_bw = -1;
    long long dc_key_delay = 0, dc_count = 0;
    // measure_audio (quisk.c:3011-3020): d_ma[0] = running sum of squares, d_ma[1] = measured_audio
    double *d_ma = nullptr; long long ma_count = 0; int ma_time = 1;
    // measure_frequency (quisk.c:3000-3009; measure_freq, quisk.c:4850-4920): three half-bands on receiver 0's decimated stream,
    // frames of 12000 samples, Hanning window, magnitude average in order of frequency, peak search around the RIT offset
    int mf_mode = 0, mf_index = 0, mf_count = 0; double mf_value = 0.0;
    Chain mf_chain; AnyFft *mf_fft = nullptr; double2 *d_mf_frame = nullptr, *d_mf_X = nullptr; double *d_mf_avg = nullptr;
    // audio graph (quisk_calc_audio_graph, quisk.c:903-964; get_audio_graph, quisk.c:966-991): frames of data_width detected
    // LSB samples of receiver 0
    int ag_index = 0, ag_count = 0, ag_ready = 0, ag_N = 0, ag_frames = 1; bool ag_inited = false;
    AnyFft *ag_fft = nullptr; double2 *d_ag_frame = nullptr, *d_ag_X = nullptr; double *d_ag_avg = nullptr;
    // test tone (AddTestTone, quisk.c:1194-1229): power tables of the tone's and the audio vector's phase steps, the two
    // vectors at the next block's first sample (ping-pong), the block with the tone in it
    int tone_freq = 0, tone_pp = 0;
    double2 *d_tone_tab = nullptr, *d_tone_state[2] = {nullptr, nullptr}, *d_tone_buf = nullptr;
    // host copy of receiver 0's band-pass taps in the reference's order (cFilterI[0] / cFilterQ[0]: get_filter reads them)
    std::vector<double> f0_i, f0_q;
    // a plain transform of any size for the module's dft / idft / get_filter (quisk.c:4761-4848, 4922-5046)
    AnyFft *gen = nullptr; int gen_n = 0; double2 *d_gen_in = nullptr, *d_gen_out = nullptr;
    // noise blanker (k_nb_*): two sets of history arrays (a block reads one and leaves the next block's history in the other)
    int nb_level = 0, nb_hw = 0, nb_H = 0, nb_set = 0;
    double2 *d_nb_x[2] = {nullptr, nullptr}, *d_nb_y = nullptr;
    double *d_nb_mag[2] = {nullptr, nullptr}, *d_nb_loc = nullptr, *d_nb_segsum = nullptr, *d_nb_segbase = nullptr, *d_nb_S = nullptr;
    unsigned char *d_nb_p[2] = {nullptr, nullptr}, *d_nb_any = nullptr;
    unsigned long long *d_nb_words = nullptr;
    double2 *d_cond = nullptr, *d_dc_seg_end = nullptr, *d_dc_seg_in = nullptr, *d_dc_state = nullptr;   // d_dc_state: [0] recurrence, [1] sum, [2] average
    double2 *d_dc_partial = nullptr;
    // output
    double2 *d_out = nullptr; size_t out_stride = 0;
    std::vector<int> out_count;
    // SSB squelch (quisk.c:1022-1116) and auto-notch (quisk.c:730-899)
    int ssb_sq_enabled = 0, ssb_sq_level = 0, rit_freq = 0;
    double *d_ssb_carry = nullptr; int *d_ssb_state = nullptr, *d_bw = nullptr, *d_notch_on = nullptr;
    NotchState *d_notch = nullptr; std::vector<int> notch_on; bool any_notch = false;
    // second-receiver routing (quisk.c:2410-2494) and Buffer2Chan's FIFOs (quisk.c:1452-1487), one per routed receiver
    // A route owns the part of the chain that the reference runs on the PAIR (quisk.c:2419-2572): Buffer2Chan's FIFOs at the
    // decimated rate, then -- `tail`, two rows: 0 = left, 1 = right -- cFracDecim, the play-rate interpolators and the two AGCs.
    // AGC slot `slot` (left) holds Agc1 while the route is live (the receiver's own state, moved in and out), slot + 1 is Agc2.
    struct Route { int rx = 0, partner = 0, method = 0, nbuf_a = 0, nbuf_b = 0, cur = 0; double *fifo[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
                   Chain tail; int slot = 0; bool live = false;
                   // mono behind a stretch of stereo: the route runs on (k_route_drain_apply) until the two channels' filter
                   // histories and Agc1's ring hold the same samples again -- `drain` output samples from the switch -- and
                   // only then hands the left channel's objects back to the receiver's class
                   bool stereo = false; long long drain = 0;
                   double *d_agc2_ring = nullptr; };     // Agc2's delay ring between two stretches of stereo
    std::vector<Route> routes; int route_cap = 0; int *d_agc_cls = nullptr, *d_zero2 = nullptr;
    // key-up envelope (quisk.c:2602-2612): keyupEnvelope and the pinned per-block factors the ramp kernel reads
    double keyup_env = 1.0; double *h_keyup = nullptr; int *h_keyup_counts = nullptr; int keyup_cap = 0;
    hipEvent_t ev_keyup = nullptr; bool keyup_used = false;
    // display spectrum (get_graph)
    int fftN = 0, fftN1 = 1, fftW = 0, fft_fill = 0, fft_count = 0;
    int fft_batch = FFT_BATCH;             // frames per launch pair (QK_FFT_BATCH, 1..FFT_BATCH)
    int fft_parts = 0;                     // partial averages of the four-step row pass that hold data (0 after a release)
    double *h_graph = nullptr;             // pinned staging of a released row + the S-meter sum
    long long graph_rows = 0;              // rows released so far (current_graph is all zero before the first)
    bool rel_pending = false;              // a row marked by qk_graph_mark (or a release under way) waits to be collected
    int rel_count = 0;                     // ... the frames in it
    hipEvent_t ev_graph = nullptr;
    std::vector<double> last_row;          // the row of the last release, on the host
    double k0_zoom = 0, k0_deltaf = 0; bool k0_valid = false;     // what d_k0 was made for
    double2 *d_fft_stage = nullptr, *d_fft_Y = nullptr, *d_tw4096 = nullptr, *d_tw_lo = nullptr, *d_tw_hi = nullptr;
    double *d_window = nullptr, *d_fft_avg = nullptr, *d_band = nullptr, *d_meter = nullptr, *d_graph = nullptr, *d_pix = nullptr;
    int *d_k0 = nullptr;
    int band_cap = 0;
    Bluestein *blue = nullptr;             // display spectrum of a size that is not a power of two
    // sub-receiver graphs (get_multirx_graph) and bandscope (get_bandscope): transforms made on first use
    AnyFft *mrx = nullptr; int mrx_width = 0;
    double2 *d_mrx_in = nullptr, *d_mrx_X = nullptr; double *d_mrx_out = nullptr;
    AnyFft *bs = nullptr; int bs_size = 0, bs_count = 0, bs_out_cap = 0;
    double *d_bs_in = nullptr, *d_bs_avg = nullptr, *d_bs_out = nullptr; double2 *d_bs_X = nullptr;
    unsigned long long *d_bs_max = nullptr;
    double smeter = -160.0 + 4.25969;
    // change_scan (quisk.c:3865-3872): frames numbered 0 .. scan_blocks-1 are stitched into d_scan_avg (order of frequency)
    int scan_blocks = 0, scan_rate = 0, scan_vfo0 = 0, scan_deltaf = 0, scan_tx_tune = 0, scan_vfo_screen = 0, scan_next = 0;
    double scan_valid = 0.84; double *d_scan_avg = nullptr;
    // raw scope (get_graph with use_fft == 0, quisk.c:4642-4649): a queue of FFT_ARRAY_SIZE = 4 frame heads (quisk.c:66)
    int raw_mode = 0, raw_head = 0, raw_count = 0;
    double2 *d_raw_rows = nullptr;
    // cRxFilterOut for callers outside the chain (qk_rx_filter_block): per bank [hist tap_cap][max_block] and an output block
    double2 *d_rfb_x[3] = {nullptr, nullptr, nullptr}, *d_rfb_y = nullptr, *d_rfb_t = nullptr;
    // RCCL communicator for the output gather (qk_comm_init): one rank per GPU, receivers sharded over the ranks
    ncclComm_t comm = nullptr; int comm_rank = 0, comm_world = 1; int *d_counts = nullptr;
    // stats / timing
    qk_stats stats{};
    unsigned timing = 0;       // bit b: time kernel group b (QK_KT_*); bit 8: the whole block
    // HIP-event timing: pairs are recorded on the engine's stream around the timed launches and read
    // back lazily (at qk_get_stats or when the pool runs out), so timing adds no host synchronisation
    std::vector<hipEvent_t> ev_pool;
    struct EvPair { int bucket; int e0, e1; };
    std::vector<EvPair> ev_pending;
    int ev_next = 0;
};

namespace {

// ---------------------------------------------------------------------------------- planning (pure)
int plan_decimation(int rate, int *d2, int *d3, int *d5)
{   // PlanDecimation, quisk.c:1509-1547
    int best = rate, b2 = 0, b3 = 0, b5 = 0;
    for (int i2 = 0; i2 <= 6; i2++)
        for (int i3 = 0; i3 <= 3; i3++)
            for (int i5 = 0; i5 <= 3; i5++) {
                int r = rate;
                for (int i = 0; i < i2; i++) r /= 2;
                for (int i = 0; i < i3; i++) r /= 3;
                for (int i = 0; i < i5; i++) r /= 5;
                if (r >= 48000 && r < best) { b2 = i2; b3 = i3; b5 = i5; best = r; }
            }
    if (best >= 50000) best = best * 24 / 25;
    if (d2) { *d2 = b2; *d3 = b3; *d5 = b5; }
    return best;
}
int rate_case_of(int rate) { return (rate + 100) / 1000; }
int decim_rate(int rate)
{
    switch (rate_case_of(rate)) {
    case 41: return 48000;
    case 53: return rate;
    case 111: case 133: return rate / 2;
    case 185: return rate / 3;
    case 370: return rate / 6;
    case 740: return rate / 12;
    case 1333: return rate / 24;
    }
    int d2, d3, d5, r = rate;
    plan_decimation(rate, &d2, &d3, &d5);
    for (int i = 0; i < d2; i++) r /= 2;
    for (int i = 0; i < d3; i++) r /= 3;
    for (int i = 0; i < d5; i++) r /= 5;
    if (r >= 50000) r = r * 24 / 25;
    return r;
}
int filter_rate(int rate, int mode, int bw)
{   // get_filter_rate, quisk.c:2661-2735 (case 41 falls through to case 53 there)
    int decim;
    switch (rate_case_of(rate)) {
    case 41: case 53: decim = rate; break;
    case 111: case 133: decim = rate / 2; break;
    case 185: decim = rate / 3; break;
    case 370: decim = rate / 6; break;
    case 740: decim = rate / 12; break;
    case 1333: decim = rate / 24; break;
    default: decim = plan_decimation(rate, nullptr, nullptr, nullptr); break;
    }
    switch (mode) {
    case QK_CWL: case QK_CWU: return decim / 8;
    case QK_AM: return decim / 2;
    case QK_FM: case QK_DGT_FM: case QK_DGT_IQ: return decim;
    case QK_DGT_U: case QK_DGT_L: return bw < 3000 ? decim / 8 : decim;
    case QK_FDV_U: case QK_FDV_L: return 8000;
    default: return decim / 4;
    }
}

int class_of_mode(int mode, int bandwidth)
{
    switch (mode) {
    case QK_CWL: case QK_CWU: return CL_CW;
    // DGT-U/L (quisk.c:1922-1973): below DGT_NARROW_FREQ (quisk.c:35) the chain is CW's (two half-bands, FIR98/2, band-pass,
    // x2, two half-band interpolators; the sideband sign comes from the mode); otherwise the band-pass runs at the decimated rate
    case QK_DGT_U: case QK_DGT_L: return bandwidth < 3000 ? CL_CW : CL_DGTW;
    case QK_AM: return CL_AM;
    case QK_FM: case QK_DGT_FM: return CL_FM;
    case QK_LSB: case QK_USB: case QK_IMD: return CL_SSB;
    case QK_DGT_IQ: return CL_DGTIQ;
    default: return -1;
    }
}

// ---------------------------------------------------------------------------------- chain building
struct ChainBuilder {
    Chain &c; int in_cap; int cur; int cur_cap;
    ChainBuilder(Chain &ch, int input_cap) : c(ch), in_cap(input_cap), cur(-1), cur_cap(input_cap) {}
    int new_stream(int planes, int cap)
    {
        Stream s; s.planes = planes; s.cap = cap + 8; s.hist = 0;
        c.streams.push_back(s);
        return (int)c.streams.size() - 1;
    }
    void need_hist(int stream, int h)   // stream -1 is the chain input, owned by the caller
    {
        if (stream >= 0) c.streams[stream].hist = std::max(c.streams[stream].hist, (h + 1) & ~1);
        else in_hist = std::max(in_hist, (h + 1) & ~1);
    }
    int in_hist = 0;
    Stage &add(StageKind k, int planes_out, int cap_out)
    {
        Stage st; st.kind = k; st.in = cur; st.out = new_stream(planes_out, cap_out);
        c.stages.push_back(st);
        cur = st.out; cur_cap = cap_out;
        return c.stages.back();
    }
    void fir(const FirTab &f, int planes, int timer)
    {
        need_hist(cur, f.D * f.TQP + f.D);      // the tiled kernel reads whole tap chunks (zero-padded taps)
        Stage &s = add(ST_FIR, planes, cur_cap / f.D + 2);
        s.h = f.hp; s.hf = f.hpf; s.T = f.T; s.D = f.D; s.TQP = f.TQP; s.timer = timer;
    }
    void interp(const double *h, int T, int L, int timer)
    {
        need_hist(cur, T / L - 1);
        Stage &s = add(ST_INTERP, 1, cur_cap * L + 2);
        s.h = h; s.T = T; s.L = L; s.timer = timer;
    }
    void hbint(int planes, int timer)
    {
        need_hist(cur, 21);
        Stage &s = add(ST_HBINT, planes, cur_cap * 2 + 2);
        s.timer = timer;
    }
    void resamp(const double *h, int T, int L, int D, int timer)
    {
        need_hist(cur, T / L - 1);
        Stage &s = add(ST_RESAMP, 2, (int)((long long)cur_cap * L / D) + 4);
        s.h = h; s.T = T; s.L = L; s.D = D; s.timer = timer;
    }
};

}  // namespace

// =================================================================================== engine methods
static void free_chain(Chain &c)
{
    for (auto &s : c.streams) if (s.d) hipFree(s.d);
    for (auto &st : c.stages) if (st.iir_state) hipFree(st.iir_state);
    if (c.d_members) hipFree(c.d_members);
    if (c.play_alt) hipFree(c.play_alt);
    if (c.front_alt) hipFree(c.front_alt);
    if (c.d_frac_idx) hipFree(c.d_frac_idx);
    if (c.d_frac_pos) hipFree(c.d_frac_pos);
    for (auto &fs : c.frac_slot) {
        if (fs.idx) hipHostFree(fs.idx);
        if (fs.pos) hipHostFree(fs.pos);
        if (fs.ev) hipEventDestroy(fs.ev);
    }
    c = Chain();
}

static int alloc_chain_streams(qk_engine *e, Chain &c, int rows = -1)
{
    for (auto &s : c.streams) {
        size_t bytes = (size_t)(rows > 0 ? rows : e->nrx) * s.rx_stride() * sizeof(double);
        HIPCHECK(hipMalloc(&s.d, bytes));
        HIPCHECK(hipMemsetAsync(s.d, 0, bytes, e->stream));
    }
    return QK_OK;
}

// Front chain: everything quisk_process_decimate does after the fused half-bands (quisk.c:1607-1720).
static int build_front(qk_engine *e)
{
    const int rate = e->cfg.sample_rate;
    int d2 = 0, d3 = 0, d5 = 0;
    e->rate_case = rate_case_of(rate);
    Chain &c = e->front;
    int S = 0;
    std::vector<const FirTab *> firs;
    bool resample = false;
    switch (e->rate_case) {
    case 41: break;
    case 53: firs.push_back(&e->f_s53); break;
    case 111: firs.push_back(&e->f_s111); break;
    case 133: firs.push_back(&e->f_s133); break;
    case 185: firs.push_back(&e->f_s185); break;
    case 370: S = 1; firs.push_back(&e->f_s185); break;
    case 740: S = 2; firs.push_back(&e->f_s185); break;
    case 1333: S = 3; firs.push_back(&e->f_s167); break;
    default: {
        plan_decimation(rate, &d2, &d3, &d5);
        S = std::min(std::max(d2 - 1, 0), 5);
        for (int i = 0; i < d3; i++) firs.push_back(&e->f_dec3);
        for (int i = 0; i < d5; i++) firs.push_back(&e->f_dec5);
        if (d2 > 0) firs.push_back(&e->f_dec2);
        int r = rate;
        for (int i = 0; i < d2; i++) r /= 2;
        for (int i = 0; i < d3; i++) r /= 3;
        for (int i = 0; i < d5; i++) r /= 5;
        resample = r >= 50000;
        break;
    }
    }
    e->S = S;
    e->decim_srate = decim_rate(rate);
    ChainBuilder b(c, e->cfg.max_block);
    // stream 0: front-end output (mixer + S half-bands)
    b.cur = b.new_stream(2, (e->cfg.max_block >> S) + 2);
    b.cur_cap = (e->cfg.max_block >> S) + 2;
    for (auto *f : firs) b.fir(*f, 2, QK_KT_DECIM);
    if (resample) {
        b.resamp(e->h_rs65, 125, 6, 5, QK_KT_DECIM);
        b.resamp(e->h_dec5, 245, 4, 5, QK_KT_DECIM);
        if (d5 == 3) {
            // The reference runs the resampler's second stage on filtDecim5S, the state object of the third /5 stage
            // (quisk.c:1701, 1713): ONE ring of 245 samples and ONE decim_index serve both, so each filter sees the other's
            // samples as its history and continues the other's counter.  Here: one stream holds, block after block, the
            // third /5 stage's input followed by the x4/5 stage's input, both stages read it, and they share `phase`.
            const int k3 = d3 + 2, kA = (int)c.stages.size() - 2, kB = kA + 1;      // third /5, x6/5, x4/5
            Stream &V = c.streams[c.stages[k3].in], &X = c.streams[c.stages[kB].in];
            V.cap += X.cap; V.hist = std::max(V.hist, X.hist);
            c.shared = c.stages[k3].in;
            c.stages[kA].out = c.shared; c.stages[kA].out_after = true;
            c.stages[kB].in = c.shared; c.stages[kB].in_after = true; c.stages[kB].phase_share = k3;
        }
    }
    c.aud = b.cur;      // "aud" of the front chain = the decimated complex stream every class reads
    c.built = true;
    return QK_OK;
}

static int build_class(qk_engine *e, int cl)
{
    Chain &c = e->cls_chain[cl];
    Stream &dec = e->front.streams[e->front.aud];
    ChainBuilder b(c, dec.cap);
    const int T_A = QK_KT_DEMOD, T_B = QK_KT_AUDIO;
    auto rxf = [&](int kind, int planes_out) {
        b.need_hist(b.cur, e->tap_cap + FT_QB);
        Stage &s = b.add(ST_RXF, planes_out, b.cur_cap + 2);
        s.rxf_kind = kind; s.timer = T_A;
    };
    auto iir = [&](int kind) -> int {
        Stage &s = b.add(ST_IIR, 1, b.cur_cap + 2);
        s.iir_kind = kind; s.timer = T_B;
        if (kind == 1) {    // de-emphasis coefficients, quisk.c:1764-1768
            const double www = tan(M_PI * 300.0 / 48000), nnn = 1.0 / (1.0 + www);
            s.a0 = www * nnn; s.a1 = s.a0; s.b1 = nnn * (www - 1.0);
        }
        HIPCHECK(hipMalloc(&s.iir_state, sizeof(double) * 2 * e->nrx));
        HIPCHECK(hipMemsetAsync(s.iir_state, 0, sizeof(double) * 2 * e->nrx, e->stream));
        return QK_OK;
    };
    int rc = QK_OK;
    switch (cl) {
    case CL_SSB:    // quisk.c:1817-1862
        b.fir(e->f_hbdec, 2, T_A);
        b.fir(e->f_dec2, 2, T_A);
        rxf(0, 1);
        c.stages.back().hook = 1; c.stages.back().hook_div = 4;
        b.need_hist(b.cur, SSQ_N + 50 / 2);
        b.interp(e->h_a24p4, 50, 2, T_B);
        c.stages.back().delayed = true;
        b.hbint(1, T_B);
        break;
    case CL_CW:     // quisk.c:1779-1816
        b.fir(e->f_hbdec, 2, T_A);
        b.fir(e->f_hbdec, 2, T_A);
        b.fir(e->f_dec2, 2, T_A);
        rxf(0, 1);
        c.stages.back().hook = 2; c.stages.back().hook_div = 8;
        b.interp(e->h_a24p4, 50, 2, T_B);
        b.hbint(1, T_B);
        b.hbint(1, T_B);
        break;
    case CL_AM:     // quisk.c:1863-1886
        b.fir(e->f_dec2, 2, T_A);
        rxf(1, 1);
        rc = iir(0);
        b.fir(e->f_a24p6, 1, T_B);
        c.stages.back().hook = 1; c.stages.back().hook_div = 2;
        b.need_hist(b.cur, SSQ_N + 24);
        b.hbint(1, T_B);
        c.stages.back().delayed = true;
        break;
    case CL_DGTW:   // quisk.c:1929-1939, 1955-1965
        rxf(0, 1);
        c.stages.back().hook = 2;
        break;
    case CL_DGTIQ:  // quisk.c:1974-1986: dRxFilterOut on the complex stream (none at 19 kHz and above), no detector
        rxf(2, 2);
        break;
    case CL_FM: {   // quisk.c:1887-1921
        rxf(2, 2);
        b.need_hist(b.cur, 1);
        Stage &s = b.add(ST_FMDISC, 1, b.cur_cap + 2);
        s.fs = e->decim_srate; s.timer = T_A;
        rc = iir(1);
        b.fir(e->f_a24p3, 1, T_B);
        b.fir(e->f_fmhp, 1, T_B);
        b.hbint(1, T_B);
        c.stages.back().hook = 2;
        break;
    }
    }
    if (rc != QK_OK) return rc;
    {   // which of the reference's objects the stages are (quisk.c:1779-1986)
        static const int ids[CL_COUNT][8] = {
            /* CL_SSB   */ {OB_HB5, OB_D48, OB_CRX, OB_NONE, OB_HB7, OB_NONE, OB_NONE, OB_NONE},
            /* CL_CW    */ {OB_HB5, OB_HB4, OB_D48, OB_CRX, OB_NONE, OB_HB6, OB_HB7, OB_NONE},
            /* CL_AM    */ {OB_D48, OB_DRX, OB_NONE, OB_NONE, OB_HB7, OB_NONE, OB_NONE, OB_NONE},
            /* CL_FM    */ {OB_DRX, OB_NONE, OB_NONE, OB_A24P3, OB_NONE, OB_HB6, OB_NONE, OB_NONE},
            /* CL_DGTW  */ {OB_CRX, OB_NONE, OB_NONE, OB_NONE, OB_NONE, OB_NONE, OB_NONE, OB_NONE},
            /* CL_DGTIQ */ {OB_DRX, OB_NONE, OB_NONE, OB_NONE, OB_NONE, OB_NONE, OB_NONE, OB_NONE}};
        for (size_t i = 0; i < c.stages.size() && i < 8; i++) c.stages[i].obj = ids[cl][i];
        // the stage whose output measure_audio sums for bank 0, and the divisor of its rate (quisk_filter_srate)
        static const int mstage[CL_COUNT][2] = {{2, 4}, {3, 8}, {2, 2}, {2, 1}, {0, 1}, {0, 1}};
        c.stages[mstage[cl][0]].measure_div = mstage[cl][1];
    }
    c.aud = b.cur;
    // tail: fractional decimator when the decimated rate is not 48000 (quisk.c:2528-2532) ...
    if (e->decim_srate != 48000) {
        b.need_hist(b.cur, 3);
        Stage &s = b.add(ST_FRAC, cl == CL_DGTIQ ? 2 : 1, b.cur_cap + 2);
        s.timer = T_B; s.obj = OB_FRAC;
        c.frac_cap = b.cur_cap + 8;
        HIPCHECK(hipMalloc(&c.d_frac_idx, sizeof(int) * c.frac_cap));
        HIPCHECK(hipMalloc(&c.d_frac_pos, sizeof(double) * c.frac_cap));
        for (auto &fs : c.frac_slot) {
            HIPCHECK(hipHostMalloc(&fs.idx, sizeof(int) * c.frac_cap, hipHostMallocMapped));
            HIPCHECK(hipHostMalloc(&fs.pos, sizeof(double) * c.frac_cap, hipHostMallocMapped));
            HIPCHECK(hipEventCreateWithFlags(&fs.ev, sync_event_flags()));
        }
        c.frac_dindex = 1.0;                                         // dindex = 1 (quisk.c:575)
    }
    // ... then interpolation to the play rate (quisk.c:2535-2553); mono, so one plane carries both channels
    for (int r = e->cfg.playback_rate / 48000, k = 0; r > 1; r /= 2, k++) {
        b.hbint(cl == CL_DGTIQ ? 2 : 1, T_B);
        c.stages.back().obj = OB_PHB0 + k;          // HalfBand7..9 of quisk_process_samples (quisk.c:2178-2180)
    }
    c.play = b.cur;
    b.need_hist(b.cur, e->cfg.playback_rate * 15 / 1000);       // AGC_DELAY, quisk.c:30, 2036
    // the class input is the front chain's output stream
    Stream &in = e->front.streams[e->front.aud];
    if (in.d == nullptr) in.hist = std::max(in.hist, b.in_hist);
    else if (in.hist < b.in_hist) return fail(QK_EINVAL, "internal: front stream history too small");
    rc = alloc_chain_streams(e, c);
    if (rc != QK_OK) return rc;
    {
        Stream &pl = c.streams[c.play];
        const size_t bytes = (size_t)e->nrx * pl.rx_stride() * sizeof(double);
        HIPCHECK(hipMalloc(&c.play_alt, bytes));
        HIPCHECK(hipMemsetAsync(c.play_alt, 0, bytes, e->stream));
    }
    if (cl == CL_FM) {
        // fm_1 starts at 10 + 0i (quisk.c:1763): it is the history sample in front of the first block
        Stream &f = c.streams[c.stages[0].out];
        std::vector<double> ten((size_t)e->nrx, 10.0);
        for (int r = 0; r < e->nrx; r++)
            HIPCHECK(hipMemcpyAsync(f.d + (long long)r * f.rx_stride() + f.hist - 1, &ten[r], sizeof(double),
                                    hipMemcpyHostToDevice, e->stream));
        HIPCHECK(hipStreamSynchronize(e->stream));
    }
    HIPCHECK(hipMalloc(&c.d_members, sizeof(int) * e->nrx));
    c.built = true;
    return QK_OK;
}

// the class whose play stream a receiver's AGC reads: its chain class, or none while a live route runs its AGC
static int upload_agc_cls(qk_engine *e)
{
    std::vector<int> ac(e->cls);
    for (auto &r : e->routes) if (r.live) ac[r.rx] = AGC_SKIP_CLASS;
    HIPCHECK(hipMemcpy(e->d_agc_cls, ac.data(), sizeof(int) * e->nrx, hipMemcpyHostToDevice));
    return QK_OK;
}
static int upload_members(qk_engine *e)
{
    { int rca = upload_agc_cls(e); if (rca != QK_OK) return rca; }
    HIPCHECK(hipMemcpyAsync(e->d_cls, e->cls.data(), sizeof(int) * e->nrx, hipMemcpyHostToDevice, e->stream));
    for (int cl = 0; cl < CL_COUNT; cl++) e->cls_chain[cl].members.clear();
    for (int r = 0; r < e->nrx; r++) e->cls_chain[e->cls[r]].members.push_back(r);
    for (int cl = 0; cl < CL_COUNT; cl++) {
        Chain &c = e->cls_chain[cl];
        if (c.members.empty()) continue;
        if (!c.built) { int rc = build_class(e, cl); if (rc != QK_OK) return rc; }
        HIPCHECK(hipMemcpyAsync(c.d_members, c.members.data(), sizeof(int) * c.members.size(),
                                hipMemcpyHostToDevice, e->stream));
        HIPCHECK(hipStreamSynchronize(e->stream));      // members vector may be rebuilt right after
    }
    return QK_OK;
}

// A receiver moves from chain class `from` to class `to` (set_rx_mode, or DGT-U/L crossing DGT_NARROW_FREQ in set_filters).
// The reference keeps ONE set of filter objects per bank whatever the mode (Storage[bank], quisk.c:1730-1747; the band-pass
// rings; HalfBand7..9, Agc1 and cFracDecim's statics of quisk_process_samples): an object that the new mode uses holds what
// the last mode that used it left there -- fresh samples if that was the old mode, stale ones from an earlier visit if not,
// zeros if never -- and its decimation counter stands where it stopped.  Here an object's contents are the history in
// front of its stage's input stream.  On leaving a class the receiver's objects are saved ([rx][2][need] per object); on
// entering one they are put in front of the new class's streams.  Objects private to one class (the audio filters, the
// one-pole states, fm_1) simply stay in that class's rows, which nothing touches while the receiver is elsewhere.
// The counters (`phase`) belong to a class, whose receivers run in lockstep: they follow the receiver when it is the only
// member of its new class (always so for the reference's single bank); a receiver that joins others takes their phase.
// All streams idle.
static double *hist_tail(qk_engine *e, Chain &c, int stream_idx, int rx, int pl, int need);
static int ring_of_class(int cl) { return (cl == CL_SSB || cl == CL_CW || cl == CL_DGTW) ? 0 : 1; }
static Stage *ring_stage(Chain &c) { for (auto &st : c.stages) if (st.kind == ST_RXF) return &st; return nullptr; }
static int eff_ntaps(const qk_engine *e, int rx)
{   // DGT-IQ is not filtered at 19 kHz and above (quisk.c:1976); N == 0 passes the samples (quisk.c:1137, 1174)
    return (e->mode[rx] == QK_DGT_IQ && e->bandwidth[rx] >= 19000) ? 0 : e->ntaps[rx];
}
// Bring the host image of the ring that receiver rx's class `cl` uses up to date: the samples it has taken in since the last
// event, written where the reference writes them (quisk.c:1139-1141, 1150).  All streams idle.
static int ring_sync(qk_engine *e, int rx, int cl)
{
    Chain &c = e->cls_chain[cl];
    Stage *st = c.built ? ring_stage(c) : nullptr;
    if (!st || e->rings.empty()) return QK_OK;
    qk_engine::RingImage &R = e->rings[(size_t)rx * 2 + ring_of_class(cl)];
    const long long m = R.active ? st->total_in - R.cnt : 0;
    R.cnt = st->total_in;
    if (m <= 0 || R.N <= 0) return QK_OK;
    const int N = R.N, L = (int)std::min<long long>(m, N);
    if (R.cell.empty()) R.cell.assign((size_t)2 * e->tap_cap, 0.0);
    std::vector<double> tail((size_t)L);
    const int p0 = R.idx >= N ? 0 : R.idx;
    for (int pl = 0; pl < 2; pl++) {
        HIPCHECK(hipMemcpy(tail.data(), hist_tail(e, c, st->in, rx, pl, L), sizeof(double) * L, hipMemcpyDeviceToHost));
        for (int j = 0; j < L; j++) {
            const long long t = m - L + j;
            R.cell[(size_t)pl * e->tap_cap + (size_t)((p0 + t) % N)] = tail[j];
        }
    }
    R.idx = (int)((p0 + m - 1) % N) + 1;
    return QK_OK;
}
// The receiver's filter length (or its class) has changed: if the ring its class uses was last run at another length, the
// kernels' plain FIR history is replaced by the one that makes them compute what the reference computes from its ring.  The
// reference writes the newest sample at p (p = indexFilter, or 0 when that is beyond the new length, quisk.c:1139-1140) and
// sums f[k] ring[(p + k) mod N] (quisk.c:1144-1149): in the kernels' order (h[0] = f[0], h[m] = f[N - m]) tap m meets cell
// (p - m) mod N, which holds x[n - m] once the ring has gone round at this length and whatever the old length left there
// until then.  So the history handed to the kernels is x'[-m] = cell[(p - m) mod N], m = 1 .. N-1.
static int ring_apply(qk_engine *e, int rx)
{
    const int cl = e->cls[rx];
    Chain &c = e->cls_chain[cl];
    Stage *st = c.built ? ring_stage(c) : nullptr;
    if (!st) return QK_OK;
    if (e->rings.empty()) e->rings.resize((size_t)e->nrx * 2);
    qk_engine::RingImage &R = e->rings[(size_t)rx * 2 + ring_of_class(cl)];
    const int N = eff_ntaps(e, rx);
    R.cnt = st->total_in;
    R.active = N > 0;
    if (N <= 0 || N == R.N) return QK_OK;
    const int p = R.idx >= N ? 0 : R.idx, H = e->tap_cap;
    std::vector<double> hist((size_t)H, 0.0);
    for (int pl = 0; pl < 2; pl++) {
        std::fill(hist.begin(), hist.end(), 0.0);
        if (!R.cell.empty())
            for (int m = 1; m < N; m++) hist[(size_t)H - m] = R.cell[(size_t)pl * H + (size_t)(((p - m) % N + N) % N)];
        HIPCHECK(hipMemcpy(hist_tail(e, c, st->in, rx, pl, H), hist.data(), sizeof(double) * H, hipMemcpyHostToDevice));
    }
    R.N = N;
    return QK_OK;
}

static int obj_need_of(const qk_engine *e, int ob)
{
    switch (ob) {
    case OB_HB5: case OB_HB4: return 44;
    case OB_D48: return 98;
    case OB_CRX: case OB_DRX: return e->tap_cap;
    case OB_HB6: case OB_HB7: case OB_PHB0: case OB_PHB1: case OB_PHB2: return 22;
    case OB_FRAC: return 4;
    default: return 0;
    }
}
// where the last `need` samples in front of the next block of stream `s` (row rx, plane pl) lie
static double *hist_tail(qk_engine *e, Chain &c, int stream_idx, int rx, int pl, int need)
{
    const bool chain_in = stream_idx < 0;
    Chain &owner = chain_in ? e->front : c;
    const int si = chain_in ? e->front.aud : stream_idx;
    Stream &s = owner.streams[si];
    double *row = s.d + (long long)rx * s.rx_stride() + (long long)pl * (s.hist + s.cap);
    // the front end's own output stream is double-buffered: the next block's head copy takes the history from behind the last
    // block in the current buffer; every other stream has had its tail moved to the front already
    if (&owner == &e->front && si == 0 && owner.front_last_n > 0) return row + s.hist + owner.front_last_n - need;
    return row + s.hist - need;
}
static int migrate_objects(qk_engine *e, int rx, Chain &c, bool save, bool adopt_phase)
{
    for (auto &st : c.stages) {
        if (st.obj < 0) continue;
        const int ob = st.obj, need = e->obj_need[ob];
        if (st.kind == ST_FIR && st.D > 1) {
            if (save) e->obj_phase[(size_t)rx * OB_COUNT + ob] = st.phase;
            else if (adopt_phase) st.phase = e->obj_phase[(size_t)rx * OB_COUNT + ob];
        }
        if (st.kind == ST_FRAC) {
            if (save) e->obj_frac[rx] = c.frac_dindex;
            else if (adopt_phase) c.frac_dindex = e->obj_frac[rx];
        }
        if (need <= 0) continue;
        const Stream &in = st.in < 0 ? e->front.streams[e->front.aud] : c.streams[st.in];
        if (in.hist < need) return fail(QK_EINVAL, "internal: stream history shorter than the object it stands for");
        for (int pl = 0; pl < 2; pl++) {
            double *obj = e->d_obj[ob] + ((size_t)rx * 2 + pl) * need;
            double *str = hist_tail(e, c, st.in, rx, pl < in.planes ? pl : 0, need);      // mono audio is both channels
            if (save) HIPCHECK(hipMemcpyAsync(obj, str, sizeof(double) * need, hipMemcpyDeviceToDevice, e->stream));
            else if (pl < in.planes) HIPCHECK(hipMemcpyAsync(str, obj, sizeof(double) * need, hipMemcpyDeviceToDevice, e->stream));
        }
    }
    return QK_OK;
}
static int migrate_receiver(qk_engine *e, int rx, int from, int to)
{
    Chain &n = e->cls_chain[to];
    if (!n.built) return QK_OK;
    if (e->obj_phase.empty()) {
        e->obj_phase.assign((size_t)e->nrx * OB_COUNT, 0);
        e->obj_frac.assign((size_t)e->nrx, 1.0);                                   // dindex = 1 (quisk.c:575)
        for (int ob = 0; ob < OB_COUNT; ob++) {
            e->obj_need[ob] = obj_need_of(e, ob);
            if (e->obj_need[ob] <= 0) continue;
            const size_t bytes = sizeof(double) * (size_t)e->nrx * 2 * e->obj_need[ob];
            HIPCHECK(hipMalloc(&e->d_obj[ob], bytes));
            HIPCHECK(hipMemsetAsync(e->d_obj[ob], 0, bytes, e->stream));           // never used: zeros (quisk.c:1751-1762)
        }
    }
    Chain &o = e->cls_chain[from];
    if (from != to && o.built) {
        int rc = ring_sync(e, rx, from);
        if (rc == QK_OK) rc = migrate_objects(e, rx, o, true, false);
        if (rc != QK_OK) return rc;
    }
    { int rc = migrate_objects(e, rx, n, false, n.members.size() == 1); if (rc != QK_OK) return rc; }
    HIPCHECK(hipStreamSynchronize(e->stream));
    { int rc = ring_apply(e, rx); if (rc != QK_OK) return rc; }
    Stream &np = n.streams[n.play];
    if (from != to && o.built && e->agc_started[rx]) {
        // Agc1's delay ring (quisk.c:2181): the 15 ms of audio in the old class's play stream
        Stream &op = o.streams[o.play];
        const int h = std::min(op.hist, np.hist);
        for (int pl = 0; pl < np.planes; pl++) {
            const int spl = pl < op.planes ? pl : 0;        // mono audio is both channels
            const double *src = o.play_alt + (long long)rx * op.rx_stride() + (long long)spl * (op.hist + op.cap) + (op.hist - h);
            for (double *base : {np.d, n.play_alt})
                HIPCHECK(hipMemcpyAsync(base + (long long)rx * np.rx_stride() + (long long)pl * (np.hist + np.cap) + (np.hist - h), src,
                                        sizeof(double) * h, hipMemcpyDeviceToDevice, e->stream));
        }
    }
    HIPCHECK(hipStreamSynchronize(e->stream));
    return QK_OK;
}

static int upload_tune(qk_engine *e, int rx)
{
    double pre, pim;
    qk_phase_step(e->tune[rx], e->cfg.sample_rate, &pre, &pim);
    std::vector<double> C(128), B(128), A((size_t)2 * e->nA);
    qk_phase_tables(pre, pim, C.data(), B.data(), A.data(), e->nA);
    const int on = e->tune[rx] != 0;
    HIPCHECK(hipMemcpyAsync(e->d_tabC + (size_t)rx * 64, C.data(), 128 * sizeof(double), hipMemcpyHostToDevice, e->stream));
    HIPCHECK(hipMemcpyAsync(e->d_tabB + (size_t)rx * 64, B.data(), 128 * sizeof(double), hipMemcpyHostToDevice, e->stream));
    HIPCHECK(hipMemcpyAsync(e->d_tabA + (size_t)rx * e->nA, A.data(), (size_t)2 * e->nA * sizeof(double), hipMemcpyHostToDevice, e->stream));
    HIPCHECK(hipMemcpyAsync(e->d_mix_on + rx, &on, sizeof(int), hipMemcpyHostToDevice, e->stream));
    HIPCHECK(hipStreamSynchronize(e->stream));
    return QK_OK;
}

static const double *push_table(DevTables &t, const double *src, int n)
{
    size_t off = t.host.size();
    t.host.insert(t.host.end(), src, src + n);
    if (t.host.size() & 1) t.host.push_back(0.0);
    return reinterpret_cast<const double *>(off);      // offset, turned into a pointer after upload
}

// taps of a decimate-by-D FIR in the layout fir_tile reads: hp[p][q] = h[p + D q], zero-padded to TQP per branch
static FirTab push_fir(DevTables &t, const double *src, int n, int D)
{
    FirTab f; f.T = n; f.D = D;
    const int TQ = (n + D - 1) / D;
    f.TQP = (TQ + FT_QB - 1) / FT_QB * FT_QB;
    size_t off = t.host.size();
    t.host.resize(off + (size_t)D * f.TQP, 0.0);
    for (int k = 0; k < n; k++) t.host[off + (size_t)(k % D) * f.TQP + k / D] = src[k];
    f.hp = reinterpret_cast<const double *>(off);
    return f;
}

static int init_spectrum(qk_engine *e);

// fp32 mode: the front chain's streams that only float kernels touch hold floats -- the packed front end's output and what
// lies between two float FIRs (at 61.44 Msps the /3 stage reads 134 MB per block instead of 268).  Not the decimated stream
// the classes read, not a stream two stages share, not behind the mixer-only front end (QK_F32_STREAMS=0: all double).
static void mark_float_streams(qk_engine *e)
{
    Chain &c = e->front;
    if (const char *v = getenv("QK_F32_STREAMS")) if (!atoi(v)) return;
    if (!e->fp32 || !e->fe_packed || e->S < 1) return;
    for (int s = 0; s < (int)c.streams.size(); s++) {
        if (s == c.aud || s == c.shared) continue;
        bool ok = true, read = false;
        for (const auto &st : c.stages) {
            if (st.out == s && !(st.kind == ST_FIR && st.hf && !st.out_after)) ok = false;
            if (st.in == s) { read = true; if (!(st.kind == ST_FIR && st.hf && !st.in_after)) ok = false; }
            if (s == 0 && st.in < 0) { read = true; if (!(st.kind == ST_FIR && st.hf && !st.in_after)) ok = false; }
        }
        c.streams[s].f32 = ok && read;
    }
}

static int init_receivers(qk_engine *e)
{
    const qk_config &cfg = e->cfg;
    int rc = build_front(e);
    if (rc != QK_OK) return rc;
    // mixer state
    e->H = (44 << e->S) + (1 << e->S) + 128;
    if (const char *v = getenv("QK_FE_SHARE")) e->fe_share = std::max(5, std::min(100, atoi(v)));
    if (const char *v = getenv("QK_FE_ROUNDS")) e->fe_rounds = std::max(1, std::min(16, atoi(v)));
    if (const char *v = getenv("QK_FE_PACKED")) e->fe_packed = atoi(v) != 0;
    e->fuse_chains = e->small_mode;
    if (const char *v = getenv("QK_FUSE")) e->fuse_chains = atoi(v) != 0;
    if (const char *v = getenv("QK_FUSE_MAX")) e->fuse_max_in = std::max(1, atoi(v));
    mark_float_streams(e);
    {
        int per_cu = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg.device) == hipSuccess) cus = prop.multiProcessorCount;
        hipError_t oe = hipErrorUnknown;
#define QK_OCC(SS) case SS: oe = e->fp32 ? (e->fe_packed ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_frontend_pk<SS>, 64, 0) \
                                                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_frontend_wave<SS, float>, 64, 0)) \
                                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_frontend_wave<SS, double>, 64, 0); break
        switch (e->S) {
        QK_OCC(1); QK_OCC(2); QK_OCC(3); QK_OCC(4); QK_OCC(5);
        default: break;
        }
#undef QK_OCC
        if (oe == hipSuccess && per_cu > 0) e->fe_slots = per_cu * cus;
    }
    e->nA = (cfg.max_block >> 12) + 2;
    const size_t nrx = e->nrx;
    for (int i = 0; i < 2; i++) {
        HIPCHECK(hipMalloc(&e->d_mixhist[i], nrx * e->H * sizeof(double2)));
        HIPCHECK(hipMemset(e->d_mixhist[i], 0, nrx * e->H * sizeof(double2)));
        HIPCHECK(hipMalloc(&e->d_v0[i], nrx * sizeof(double2)));
    }
    std::vector<double2> ones(nrx, make_double2(1.0, 0.0));     // rxTuneVector = 1 (quisk.c:2168)
    HIPCHECK(hipMemcpy(e->d_v0[0], ones.data(), nrx * sizeof(double2), hipMemcpyHostToDevice));
    HIPCHECK(hipMalloc(&e->d_tabA, nrx * e->nA * sizeof(double2)));
    HIPCHECK(hipMalloc(&e->d_tabB, nrx * 64 * sizeof(double2)));
    HIPCHECK(hipMalloc(&e->d_tabC, nrx * 64 * sizeof(double2)));
    HIPCHECK(hipMalloc(&e->d_mix_on, nrx * sizeof(int)));
    HIPCHECK(hipMemset(e->d_mix_on, 0, nrx * sizeof(int)));
    e->tune.assign(nrx, 0); e->mode.assign(nrx, QK_USB); e->cls.assign(nrx, CL_SSB);
    e->ntaps.assign(nrx, 0); e->bandwidth.assign(nrx, 0); e->start_offset.assign(nrx, 0);
    e->out_count.assign(nrx, 0);
    for (size_t r = 0; r < nrx; r++) { rc = upload_tune(e, (int)r); if (rc != QK_OK) return rc; }
    // per-receiver parameters
    HIPCHECK(hipMalloc(&e->d_taps, nrx * 2 * e->tap_cap * sizeof(double)));
    HIPCHECK(hipMemset(e->d_taps, 0, nrx * 2 * e->tap_cap * sizeof(double)));
    if (e->fp32) {
        HIPCHECK(hipMalloc(&e->d_taps_f, nrx * 2 * e->tap_cap * sizeof(float)));
        HIPCHECK(hipMemset(e->d_taps_f, 0, nrx * 2 * e->tap_cap * sizeof(float)));
    }
    HIPCHECK(hipMalloc(&e->d_ntaps, nrx * sizeof(int)));
    HIPCHECK(hipMemset(e->d_ntaps, 0, nrx * sizeof(int)));
    HIPCHECK(hipMalloc(&e->d_cls, nrx * sizeof(int)));
    const size_t nslots = nrx + 2 * QK_MAX_ROUTES;               // the receivers' AGCs, then Agc1 / Agc2 of every stereo route
    HIPCHECK(hipMalloc(&e->d_mode, nslots * sizeof(int)));
    HIPCHECK(hipMemset(e->d_mode, 0, nslots * sizeof(int)));
    HIPCHECK(hipMemcpy(e->d_mode, e->mode.data(), nrx * sizeof(int), hipMemcpyHostToDevice));
    HIPCHECK(hipMalloc(&e->d_agc_cls, nrx * sizeof(int)));
    HIPCHECK(hipMalloc(&e->d_zero2, 2 * sizeof(int)));
    HIPCHECK(hipMemset(e->d_zero2, 0, 2 * sizeof(int)));
    std::vector<double> rel(nslots, 80.0);                       // agcReleaseGain, quisk.c:145
    HIPCHECK(hipMalloc(&e->d_release_gain, nslots * sizeof(double)));
    HIPCHECK(hipMemcpy(e->d_release_gain, rel.data(), nslots * sizeof(double), hipMemcpyHostToDevice));
    // AGC state as process_agc initialises it (quisk.c:2038-2044)
    std::vector<AgcState> ag(nslots);
    for (auto &a : ag) { a.themax = 1.0; a.gain = 100; a.delta = 0; a.target_gain = 100; a.index_read = a.index_start = a.is_clipping = a.pad = 0; }
    HIPCHECK(hipMalloc(&e->d_agc, nslots * sizeof(AgcState)));
    HIPCHECK(hipMemcpy(e->d_agc, ag.data(), nslots * sizeof(AgcState), hipMemcpyHostToDevice));
    HIPCHECK(hipMalloc(&e->d_agc_started, nslots * sizeof(int)));
    HIPCHECK(hipMemset(e->d_agc_started, 0, nslots * sizeof(int)));
    e->agc_started.assign(nslots, 0);
    HIPCHECK(hipHostMalloc(&e->h_agc_abort, sizeof(int), hipHostMallocMapped));
    *e->h_agc_abort = 0;
    HIPCHECK(hipHostGetDevicePointer((void **)&e->d_agc_abort, e->h_agc_abort, 0));
    // output: the audio of one block at the play rate
    // (a stereo route's block can be longer than a receiver's by what its FIFOs held: BUF2CHAN_SIZE at the most, quisk.c:1452)
    e->route_cap = 12000;
    const long long per_block = (long long)cfg.max_block * cfg.playback_rate / cfg.sample_rate + 64 +
                                (long long)e->route_cap * (cfg.playback_rate / 48000) * 5 / 4;
    if (per_block > (1LL << 26)) return fail(QK_ERANGE, "max_block gives more than 2^26 audio samples per block (k_agc_gain tile index)");
    e->out_stride = (size_t)((per_block + 63) & ~63LL);
    e->gain_stride = (int)e->out_stride;
    if (e->small_mode) {
        HIPCHECK(hipHostMalloc(&e->d_out, nslots * e->out_stride * sizeof(double2), hipHostMallocDefault));
        memset(e->d_out, 0, nslots * e->out_stride * sizeof(double2));
    } else {
        HIPCHECK(hipMalloc(&e->d_out, nslots * e->out_stride * sizeof(double2)));
        HIPCHECK(hipMemset(e->d_out, 0, nslots * e->out_stride * sizeof(double2)));
    }
    HIPCHECK(hipMalloc(&e->d_ssb_carry, nrx * SSQ_N * sizeof(double)));
    HIPCHECK(hipMemset(e->d_ssb_carry, 0, nrx * SSQ_N * sizeof(double)));
    HIPCHECK(hipMalloc(&e->d_ssb_state, nrx * 4 * sizeof(int)));
    HIPCHECK(hipMemset(e->d_ssb_state, 0, nrx * 4 * sizeof(int)));
    HIPCHECK(hipMalloc(&e->d_bw, nrx * sizeof(int)));
    HIPCHECK(hipMemset(e->d_bw, 0, nrx * sizeof(int)));
    HIPCHECK(hipMalloc(&e->d_notch_on, nrx * sizeof(int)));
    HIPCHECK(hipMemset(e->d_notch_on, 0, nrx * sizeof(int)));
    e->notch_on.assign(nrx, 0);
    e->keyup_cap = (int)(cfg.playback_rate * 5e-3) + 8;
    HIPCHECK(hipHostMalloc(&e->h_keyup, sizeof(double) * e->keyup_cap, hipHostMallocMapped));
    HIPCHECK(hipHostMalloc(&e->h_keyup_counts, sizeof(int) * nrx, hipHostMallocMapped));
    HIPCHECK(hipEventCreateWithFlags(&e->ev_keyup, sync_event_flags()));
    HIPCHECK(hipMalloc(&e->d_gains, nslots * e->gain_stride * sizeof(double)));
    HIPCHECK(hipMalloc(&e->d_agc_mag, nrx * e->gain_stride * sizeof(double)));
    if (e->apply_stream) {
        HIPCHECK(hipMalloc(&e->d_gains2, nrx * e->gain_stride * sizeof(double)));
        HIPCHECK(hipMalloc(&e->d_agc_started2, nrx * sizeof(int)));
        HIPCHECK(hipMemset(e->d_agc_started2, 0, nrx * sizeof(int)));
    }
    HIPCHECK(hipMalloc(&e->d_ma, 2 * sizeof(double)));
    HIPCHECK(hipMemset(e->d_ma, 0, 2 * sizeof(double)));
    HIPCHECK(hipMalloc(&e->d_sq_state, nrx * 3 * sizeof(double)));
    HIPCHECK(hipMemset(e->d_sq_state, 0, nrx * 3 * sizeof(double)));
    e->sq_pstride = cfg.max_block / 256 + 8;        // the discriminator's workgroups: its stream is never longer than the input block
    HIPCHECK(hipMalloc(&e->d_sq_partial, (size_t)nrx * e->sq_pstride * sizeof(double)));
    HIPCHECK(hipMalloc(&e->d_sq_flag, 2 * nrx * sizeof(int)));
    HIPCHECK(hipMemset(e->d_sq_flag, 0, 2 * nrx * sizeof(int)));
    // class chains are built when a receiver first enters the class (all receivers start as USB).  Their
    // input is the front chain's last stream: give it the largest history any class reads back
    // (the receive band-pass of the FM class reads it directly; the others put a 98-tap FIR first).
    {
        Stream &dec = e->front.streams[e->front.aud];
        dec.hist = std::max(dec.hist, std::max(e->tap_cap + FT_QB, 2 * 56 + 2));
    }
    rc = alloc_chain_streams(e, e->front);
    if (rc != QK_OK) return rc;
    {
        const size_t bytes = (size_t)e->nrx * e->front.streams[0].rx_stride() * sizeof(double);
        HIPCHECK(hipMalloc(&e->front.front_alt, bytes));
        HIPCHECK(hipMemsetAsync(e->front.front_alt, 0, bytes, e->stream));
    }
    {   // the band-pass kernel stages FT_NTH * FT_R + taps doubles in dynamic LDS: above 64 KiB the launch needs the attribute
        const size_t lds = sizeof(double) * (FT_NTH * FT_R + e->tap_cap + FT_QB);
        if (lds > 160 * 1024) return fail(QK_ERANGE, "receive-filter capacity needs more LDS than a CU has");
        if (lds > 48 * 1024) {
            HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rxf_tiled<FT_NTH, FT_R, double>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rxf_tiled<FT_NTH, FT_R, float>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
    }
    rc = upload_members(e);
    if (rc != QK_OK) return rc;
    return QK_OK;
}

static int engine_init(qk_engine *e)
{
    const qk_config &cfg = e->cfg;
    HIPCHECK(hipSetDevice(cfg.device));
    e->small_mode = cfg.reserved[2] == 1 || (cfg.reserved[2] == 0 && cfg.max_block <= 4096);
    if (const char *v = getenv("QK_SMALL_MODE")) e->small_mode = atoi(v) != 0;
    if (const char *v = getenv("QK_TIMELINE")) {
        const char *w = getenv("QK_TIMELINE_FIRST"), *c = getenv("QK_TIMELINE_BLOCKS");
        e->tl_path = v; e->tl_first = w ? atoi(w) : 60; e->tl_n = std::min(256, std::max(1, c ? atoi(c) : 48));
        e->tl_ev.resize((size_t)e->tl_n * 8);
        for (auto &ev : e->tl_ev) HIPCHECK(hipEventCreate(&ev));
    }
    HIPCHECK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    if (e->small_mode) {
        // one stream under four names: the events between the stages become same-stream no-ops
        e->agc_stream = e->apply_stream = e->fe_stream = e->stream;
        HIPCHECK(hipEventCreateWithFlags(&e->ev_gain, sync_event_flags()));
    } else {
        HIPCHECK(hipStreamCreateWithFlags(&e->agc_stream, hipStreamNonBlocking));
        bool apply_own = true;      // (QK_APPLY_STREAM=0: the limiter pass stays on the AGC's stream)
        if (const char *v = getenv("QK_APPLY_STREAM")) apply_own = atoi(v) != 0;
        if (apply_own) {
            HIPCHECK(hipStreamCreateWithFlags(&e->apply_stream, hipStreamNonBlocking));
            HIPCHECK(hipEventCreateWithFlags(&e->ev_gain, sync_event_flags()));
        }
        HIPCHECK(hipStreamCreateWithFlags(&e->fe_stream, hipStreamNonBlocking));
    }
    HIPCHECK(hipEventCreateWithFlags(&e->ev_audio, sync_event_flags()));
    HIPCHECK(hipEventCreateWithFlags(&e->ev_fe, sync_event_flags()));
    HIPCHECK(hipEventCreateWithFlags(&e->ev_fe2, sync_event_flags()));
    HIPCHECK(hipEventCreateWithFlags(&e->ev_read[0], sync_event_flags()));
    HIPCHECK(hipEventCreateWithFlags(&e->ev_read[1], sync_event_flags()));
    HIPCHECK(hipEventCreateWithFlags(&e->ev_chain, sync_event_flags()));
    HIPCHECK(hipEventCreateWithFlags(&e->ev_front, sync_event_flags()));
    HIPCHECK(hipEventCreateWithFlags(&e->ev_lane[0], sync_event_flags()));
    HIPCHECK(hipEventCreateWithFlags(&e->ev_lane[1], sync_event_flags()));
    HIPCHECK(hipEventCreateWithFlags(&e->ev_agc[0], sync_event_flags()));
    HIPCHECK(hipEventCreateWithFlags(&e->ev_agc[1], sync_event_flags()));
    e->nrx = cfg.n_receivers;
    // coefficient tables
    DevTables T;
    double hb43[43] = {0};
    static const double c12[12] = { HB_C0, HB_C1, HB_C2, HB_C3, HB_C4, HB_C5, HB_C6, HB_C7, HB_C8, HB_C9, HB_C10, HB_C11 };
    for (int k = 0; k < 11; k++) { hb43[2 * k] = c12[k]; hb43[42 - 2 * k] = c12[k]; }
    hb43[21] = c12[11];
    const double *o_dec5 = push_table(T, qk_h_dec5_245, 245), *o_rs65 = push_table(T, qk_h_rs65_125, 125),
                 *o_a4 = push_table(T, qk_h_aud24p4_50, 50);
    e->f_dec2 = push_fir(T, qk_h_dec2_98, 98, 2);        e->f_dec3 = push_fir(T, qk_h_dec3_147, 147, 3);
    e->f_dec5 = push_fir(T, qk_h_dec5_245, 245, 5);      e->f_hbdec = push_fir(T, hb43, 43, 2);
    e->f_s53 = push_fir(T, qk_h_sdriq53_55, 55, 1);      e->f_s111 = push_fir(T, qk_h_sdriq111_114, 114, 2);
    e->f_s133 = push_fir(T, qk_h_sdriq133_136, 136, 2);  e->f_s167 = push_fir(T, qk_h_sdriq167_174, 174, 3);
    e->f_s185 = push_fir(T, qk_h_sdriq185_189, 189, 3);  e->f_a24p3 = push_fir(T, qk_h_aud24p3_100, 100, 2);
    e->f_a24p6 = push_fir(T, qk_h_aud24p6_36, 36, 1);    e->f_fmhp = push_fir(T, qk_h_fmhp_309, 309, 1);
    HIPCHECK(hipMalloc(&e->d_coefs, T.host.size() * sizeof(double)));
    HIPCHECK(hipMemcpy(e->d_coefs, T.host.data(), T.host.size() * sizeof(double), hipMemcpyHostToDevice));
#define FIX(dst, off) e->dst = e->d_coefs + reinterpret_cast<size_t>(off)
    FIX(h_dec5, o_dec5); FIX(h_rs65, o_rs65); FIX(h_a24p4, o_a4);
    if (e->fp32) {
        std::vector<float> tf(T.host.begin(), T.host.end());
        HIPCHECK(hipMalloc(&e->d_coefs_f, tf.size() * sizeof(float)));
        HIPCHECK(hipMemcpy(e->d_coefs_f, tf.data(), tf.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    for (FirTab *f : {&e->f_dec2, &e->f_dec3, &e->f_dec5, &e->f_hbdec, &e->f_s53, &e->f_s111, &e->f_s133, &e->f_s167,
                      &e->f_s185, &e->f_a24p3, &e->f_a24p6, &e->f_fmhp}) {
        if (e->fp32) f->hpf = e->d_coefs_f + reinterpret_cast<size_t>(f->hp);
        f->hp = e->d_coefs + reinterpret_cast<size_t>(f->hp);
    }
#undef FIX
    int rc = QK_OK;
    {
        std::vector<double2> tw(4096);
        for (int j = 0; j < 4096; j++) tw[j] = make_double2(cos(2.0 * M_PI * j / 4096), -sin(2.0 * M_PI * j / 4096));
        HIPCHECK(hipMalloc(&e->d_tw4096, sizeof(double2) * 4096));
        HIPCHECK(hipMemcpy(e->d_tw4096, tw.data(), sizeof(double2) * 4096, hipMemcpyHostToDevice));
    }
    if (e->small_mode) {
        // the two input blocks: fine-grained DEVICE memory the host stores into directly over the bus (large BAR; 0.5 us per
        // 960-sample block, write-combined) -- the front end then reads local memory; where the platform refuses that, pinned
        // host memory the kernels read over the bus (the front end's launch is 8 us longer)
        const size_t ib = (size_t)std::max(cfg.max_block, 1) * sizeof(double2);
        int large_bar = 0;          // (the whole of device memory is mapped for the host only with a large BAR)
        if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, cfg.device) != hipSuccess) large_bar = 0;
        e->in_on_device = large_bar && hipExtMallocWithFlags((void **)&e->d_in, ib, hipDeviceMallocFinegrained) == hipSuccess &&
                          hipExtMallocWithFlags((void **)&e->d_in_alt, ib, hipDeviceMallocFinegrained) == hipSuccess;
        if (const char *v = getenv("QK_IN_ON_DEVICE")) if (!atoi(v) && e->in_on_device) { hipFree(e->d_in); hipFree(e->d_in_alt); e->d_in = e->d_in_alt = nullptr; e->in_on_device = false; }
        if (!e->in_on_device) {
            (void)hipGetLastError();
            if (e->d_in) { hipFree(e->d_in); e->d_in = nullptr; }
            HIPCHECK(hipHostMalloc(&e->d_in, ib, hipHostMallocDefault));
            HIPCHECK(hipHostMalloc(&e->d_in_alt, ib, hipHostMallocDefault));
        }
        for (auto &ev : e->ev_in) HIPCHECK(hipEventCreateWithFlags(&ev, sync_event_flags()));
    }
    else HIPCHECK(hipMalloc(&e->d_in, (size_t)cfg.max_block * sizeof(double2)));
    {
        const int nseg = (cfg.max_block + DC_SEG - 1) / DC_SEG;
        HIPCHECK(hipMalloc(&e->d_cond, (size_t)cfg.max_block * sizeof(double2)));
        HIPCHECK(hipMalloc(&e->d_dc_seg_end, (size_t)nseg * sizeof(double2)));
        HIPCHECK(hipMalloc(&e->d_dc_seg_in, (size_t)nseg * sizeof(double2)));
        HIPCHECK(hipMalloc(&e->d_dc_state, 3 * sizeof(double2)));
        HIPCHECK(hipMemset(e->d_dc_state, 0, 3 * sizeof(double2)));
        HIPCHECK(hipMalloc(&e->d_dc_partial, 256 * sizeof(double2)));
    }
    if (e->nrx > 0) { rc = init_receivers(e); if (rc != QK_OK) return rc; }
    if (cfg.fft_size > 0) { rc = init_spectrum(e); if (rc != QK_OK) return rc; }
    e->ev_pool.resize(EV_POOL);
    for (auto &ev : e->ev_pool) HIPCHECK(hipEventCreate(&ev));
    HIPCHECK(hipDeviceSynchronize());
    return QK_OK;
}

// ---------------------------------------------------------------------------------- per-block work
namespace {
int drain_events(qk_engine *e)
{
    if (e->ev_pending.empty()) { e->ev_next = 0; return QK_OK; }
    HIPCHECK(hipStreamSynchronize(e->fe_stream));
    HIPCHECK(hipStreamSynchronize(e->stream));
    HIPCHECK(hipStreamSynchronize(e->agc_stream));
    if (e->apply_stream) HIPCHECK(hipStreamSynchronize(e->apply_stream));      // (a lane of class chains may carry timed kernels)
    for (auto &p : e->ev_pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e->ev_pool[p.e0], e->ev_pool[p.e1]) != hipSuccess) continue;
        if (p.bucket == 8) { e->stats.block_ms_sum += ms; e->stats.blocks_timed += 1; e->stats.last_block_ms = ms; }
        else { e->stats.kernel_ms_sum[p.bucket] += ms; e->stats.kernel_launches[p.bucket] += 1; }
    }
    e->ev_pending.clear();
    e->ev_next = 0;
    return QK_OK;
}
struct Timer {      // bucket 0..7 = QK_KT_*, 8 = whole block
    qk_engine *e; int bucket; int e0 = -1; hipStream_t st;
    Timer(qk_engine *en, int b, hipStream_t stream = nullptr) : e(en), bucket(b), st(stream ? stream : en->stream)
    {
        if (b < 0 || b > 8 || !(e->timing >> b & 1u)) return;
        if (e->ev_next + 2 > (int)e->ev_pool.size()) drain_events(e);
        e0 = e->ev_next; e->ev_next += 2;
        hipEventRecord(e->ev_pool[e0], st);
    }
    ~Timer()
    {
        if (e0 < 0) return;
        hipEventRecord(e->ev_pool[e0 + 1], st);
        e->ev_pending.push_back({bucket, e0, e0 + 1});
    }
};
inline void tl_mark(qk_engine *e, int point, hipStream_t st)
{
    if (e->tl_ev.empty()) return;
    const long long b = e->pipe_blocks - e->tl_first;
    if (b < 0 || b >= e->tl_n) return;
    hipEventRecord(e->tl_ev[(size_t)b * 8 + point], st);
}
void tl_dump(qk_engine *e)
{
    if (e->tl_ev.empty()) return;
    FILE *f = fopen(e->tl_path.c_str(), "w");
    if (f) {
        fprintf(f, "block,fe_start,fe_end,chain_start,chain_end,agc_start,gain_end,agc_end,block_end\n");
        for (int b = 0; b < e->tl_n && e->tl_first + b < e->pipe_blocks; b++) {
            fprintf(f, "%d", e->tl_first + b);
            for (int k = 0; k < 8; k++) {
                float ms = -1;
                if (hipEventElapsedTime(&ms, e->tl_ev[0], e->tl_ev[(size_t)b * 8 + k]) != hipSuccess) ms = -1;
                fprintf(f, ",%.4f", ms);
            }
            fprintf(f, "\n");
        }
        fclose(f);
    }
    for (auto ev : e->tl_ev) hipEventDestroy(ev);
    e->tl_ev.clear();
}
}  // namespace

static inline dim3 grid1(int n, int planes, int nz) { return dim3((unsigned)((n + 255) / 256), (unsigned)planes, (unsigned)nz); }


// ---------------------------------------------------------------------------------- generic FFT plans
struct FftPlan {       // power-of-two complex FFT of M points in HBM (k_gfft_*)
    int M = 0, N1 = 1;
    double2 *tw_lo = nullptr, *tw_hi = nullptr, *Y = nullptr;
};
struct Bluestein {     // any length N through two FFTs of M >= 2N - 1 points
    int N = 0, M = 0;
    FftPlan plan;
    double2 *pre = nullptr;     // window[n] * chirp[n]  (or the chirp alone)
    double2 *chirp = nullptr;   // c[k] = exp(-i pi k^2 / N), k < N
    double2 *Bhat = nullptr;    // FFT_M of conj(c) wrapped
    double2 *work = nullptr;    // M
};
static bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

static void plan_free(FftPlan &p)
{
    if (p.tw_lo) hipFree(p.tw_lo);
    if (p.tw_hi) hipFree(p.tw_hi);
    if (p.Y) hipFree(p.Y);
    p = FftPlan();
}
static int plan_init(FftPlan &p, int M)
{
    if (!is_pow2(M) || M < 64 || M > (1 << 24)) return fail(QK_EUNSUPPORTED, "FFT plan: size out of range");
    p.M = M; p.N1 = M > 4096 ? M / 4096 : 1;
    if (p.N1 > 1) {
        std::vector<double2> lo(1024), hi(std::max(1, M / 1024));
        for (int j = 0; j < 1024; j++) lo[j] = make_double2(cos(2.0 * M_PI * j / M), -sin(2.0 * M_PI * j / M));
        for (size_t j = 0; j < hi.size(); j++) {
            const double t = 2.0 * M_PI * (double)j * 1024.0 / M;
            hi[j] = make_double2(cos(t), -sin(t));
        }
        HIPCHECK(hipMalloc(&p.tw_lo, sizeof(double2) * lo.size()));
        HIPCHECK(hipMemcpy(p.tw_lo, lo.data(), sizeof(double2) * lo.size(), hipMemcpyHostToDevice));
        HIPCHECK(hipMalloc(&p.tw_hi, sizeof(double2) * hi.size()));
        HIPCHECK(hipMemcpy(p.tw_hi, hi.data(), sizeof(double2) * hi.size(), hipMemcpyHostToDevice));
        HIPCHECK(hipMalloc(&p.Y, sizeof(double2) * M));
    }
    return QK_OK;
}
// out[k] (k < n_out) = FFT_M or M-scaled inverse of in[e] (e < n_in, zero beyond), with the fused element-wise steps
static void gfft_run(qk_engine *e, const FftPlan &p, hipStream_t st, const double2 *in, const double *in_real, int n_in,
                     double2 *out, int n_out, const double2 *pre, const double2 *post, bool inverse)
{
    GfftArgs a{};
    a.in = in; a.in_real = in_real; a.out = out; a.Y = p.Y; a.pre = pre; a.post = post;
    a.tw4096 = e->d_tw4096; a.tw_lo = p.tw_lo; a.tw_hi = p.tw_hi;
    a.M = p.M; a.N1 = p.N1; a.n_in = n_in; a.n_out = n_out; a.inverse = inverse ? 1 : 0; a.scale = 1.0 / p.M;
    if (p.N1 > 1) {
        hipLaunchKernelGGL(k_gfft_cols, dim3(p.N1), dim3(FFT_BIG_THREADS), 0, st, a);
        hipLaunchKernelGGL(k_gfft_rows, dim3(p.N1), dim3(FFT_BIG_THREADS), 0, st, a);
        return;
    }
    switch (p.M) {
    case 64: hipLaunchKernelGGL(k_gfft_small<64>, dim3(1), dim3(FFT_THREADS), 0, st, a); break;
    case 128: hipLaunchKernelGGL(k_gfft_small<128>, dim3(1), dim3(FFT_THREADS), 0, st, a); break;
    case 256: hipLaunchKernelGGL(k_gfft_small<256>, dim3(1), dim3(FFT_THREADS), 0, st, a); break;
    case 512: hipLaunchKernelGGL(k_gfft_small<512>, dim3(1), dim3(FFT_THREADS), 0, st, a); break;
    case 1024: hipLaunchKernelGGL(k_gfft_small<1024>, dim3(1), dim3(FFT_THREADS), 0, st, a); break;
    case 2048: hipLaunchKernelGGL(k_gfft_small<2048>, dim3(1), dim3(FFT_THREADS), 0, st, a); break;
    default: hipLaunchKernelGGL(k_gfft_small<4096>, dim3(1), dim3(FFT_THREADS), 0, st, a); break;
    }
}

static void blue_free(Bluestein *&b)
{
    if (!b) return;
    plan_free(b->plan);
    for (void *p : {(void *)b->pre, (void *)b->chirp, (void *)b->Bhat, (void *)b->work}) if (p) hipFree(p);
    delete b; b = nullptr;
}
// window: N doubles on the host, or nullptr for none
static int blue_init(qk_engine *e, Bluestein *&out, int N, const double *window)
{
    if (N < 2 || N > (1 << 23)) return fail(QK_EUNSUPPORTED, "FFT size out of range");
    Bluestein *b = new (std::nothrow) Bluestein();
    if (!b) return fail(QK_ENOMEM, "out of host memory");
    out = b;
    b->N = N;
    int M = 64;
    while (M < 2 * N - 1) M <<= 1;
    b->M = M;
    int rc = plan_init(b->plan, M);
    if (rc != QK_OK) return rc;
    std::vector<double2> c(N), pre(N), bw((size_t)M, make_double2(0.0, 0.0));
    for (long long n = 0; n < N; n++) {
        const long long r = (n * n) % (2LL * N);                 // the phase pi n^2 / N, reduced exactly
        const double ang = M_PI * (double)r / (double)N;
        c[n] = make_double2(cos(ang), -sin(ang));
        const double w = window ? window[n] : 1.0;
        pre[n] = make_double2(c[n].x * w, c[n].y * w);
        const double2 cc = make_double2(c[n].x, -c[n].y);        // conj(c[n]) at lags +n and -n
        bw[n] = cc;
        if (n) bw[M - n] = cc;
    }
    HIPCHECK(hipMalloc(&b->chirp, sizeof(double2) * N));
    HIPCHECK(hipMemcpy(b->chirp, c.data(), sizeof(double2) * N, hipMemcpyHostToDevice));
    HIPCHECK(hipMalloc(&b->pre, sizeof(double2) * N));
    HIPCHECK(hipMemcpy(b->pre, pre.data(), sizeof(double2) * N, hipMemcpyHostToDevice));
    HIPCHECK(hipMalloc(&b->Bhat, sizeof(double2) * M));
    HIPCHECK(hipMalloc(&b->work, sizeof(double2) * M));
    HIPCHECK(hipMemcpy(b->work, bw.data(), sizeof(double2) * M, hipMemcpyHostToDevice));
    gfft_run(e, b->plan, e->fe_stream, b->work, nullptr, M, b->Bhat, M, nullptr, nullptr, false);
    HIPCHECK(hipStreamSynchronize(e->fe_stream));
    return QK_OK;
}
// X[k], k < N, natural order, of x[n] (complex, or real when x_real is given) times the plan's window
static void blue_run(qk_engine *e, Bluestein *b, hipStream_t st, const double2 *x, const double *x_real, double2 *X)
{
    gfft_run(e, b->plan, st, x, x_real, b->N, b->work, b->M, b->pre, b->Bhat, false);
    gfft_run(e, b->plan, st, b->work, nullptr, b->M, X, b->N, nullptr, b->chirp, true);
}

// a windowed forward transform of any length: the plain plan for powers of two, Bluestein otherwise
struct AnyFft {
    int N = 0;
    FftPlan plan; bool use_plan = false;
    Bluestein *blue = nullptr;
    double2 *winc = nullptr;    // the window as complex numbers (plan path: fused into the load)
};
static void any_free(AnyFft *&f)
{
    if (!f) return;
    plan_free(f->plan);
    blue_free(f->blue);
    if (f->winc) hipFree(f->winc);
    delete f; f = nullptr;
}
static int any_init(qk_engine *e, AnyFft *&out, int N, const double *window)
{
    AnyFft *f = new (std::nothrow) AnyFft();
    if (!f) return fail(QK_ENOMEM, "out of host memory");
    out = f;
    f->N = N;
    if (is_pow2(N) && N >= 64) {
        f->use_plan = true;
        int rc = plan_init(f->plan, N);
        if (rc != QK_OK) return rc;
        std::vector<double2> wc(N);
        for (int i = 0; i < N; i++) wc[i] = make_double2(window ? window[i] : 1.0, 0.0);
        HIPCHECK(hipMalloc(&f->winc, sizeof(double2) * N));
        HIPCHECK(hipMemcpy(f->winc, wc.data(), sizeof(double2) * N, hipMemcpyHostToDevice));
        return QK_OK;
    }
    return blue_init(e, f->blue, N, window);
}
static void any_run(qk_engine *e, AnyFft *f, hipStream_t st, const double2 *x, const double *x_real, double2 *X)
{
    if (f->use_plan) gfft_run(e, f->plan, st, x, x_real, f->N, X, f->N, f->winc, nullptr, false);
    else blue_run(e, f->blue, st, x, x_real, X);
}

// ---------------------------------------------------------------------------------- display spectrum
static int init_spectrum(qk_engine *e)
{
    const int N = e->cfg.fft_size, W = e->cfg.data_width;
    if (N < 2 || N > (1 << 24)) return fail(QK_EUNSUPPORTED, "fft_size out of range (2 ... 2^24)");
    if (W < 1 || W > N) return fail(QK_EINVAL, "data_width must be in [1, fft_size]");
    const bool direct = is_pow2(N) && N >= 64;          // anything else goes through Bluestein (quisk.py:3858-3890 sizes)
    if (const char *v = getenv("QK_FFT_BATCH")) e->fft_batch = std::max(1, std::min(FFT_BATCH, atoi(v)));
    e->fftN = N; e->fftW = W; e->fftN1 = direct && N > 4096 ? N / 4096 : 1;
    std::vector<double> win(N);
    for (int i = 0, j = -N / 2; i < N; i++, j++) win[i] = 0.5 + 0.5 * cos(2. * M_PI * j / N);   // quisk.c:5091-5096
    HIPCHECK(hipMalloc(&e->d_window, sizeof(double) * N));
    HIPCHECK(hipMemcpy(e->d_window, win.data(), sizeof(double) * N, hipMemcpyHostToDevice));
    if (direct) {
        std::vector<double2> lo(1024), hi(std::max(1, N / 1024));
        for (int j = 0; j < 1024; j++) lo[j] = make_double2(cos(2.0 * M_PI * j / N), -sin(2.0 * M_PI * j / N));
        for (size_t j = 0; j < hi.size(); j++) {
            const double t = 2.0 * M_PI * (double)j * 1024.0 / N;
            hi[j] = make_double2(cos(t), -sin(t));
        }
        HIPCHECK(hipMalloc(&e->d_tw_lo, sizeof(double2) * 1024));
        HIPCHECK(hipMemcpy(e->d_tw_lo, lo.data(), sizeof(double2) * 1024, hipMemcpyHostToDevice));
        HIPCHECK(hipMalloc(&e->d_tw_hi, sizeof(double2) * hi.size()));
        HIPCHECK(hipMemcpy(e->d_tw_hi, hi.data(), sizeof(double2) * hi.size(), hipMemcpyHostToDevice));
    } else {
        int rcb = blue_init(e, e->blue, N, win.data());
        if (rcb != QK_OK) return rcb;
    }
    HIPCHECK(hipMalloc(&e->d_fft_stage, sizeof(double2) * N));
    if (e->blue) HIPCHECK(hipMalloc(&e->d_fft_Y, sizeof(double2) * N));
    else if (e->fftN1 > 1) HIPCHECK(hipMalloc(&e->d_fft_Y, sizeof(double2) * N * FFT_BATCH));      // Y of up to FFT_BATCH frames per launch
    const int parts = e->fftN1 > 1 ? FFT_GROUPS : 1;
    HIPCHECK(hipMalloc(&e->d_fft_avg, sizeof(double) * N * parts));
    HIPCHECK(hipMemset(e->d_fft_avg, 0, sizeof(double) * N * parts));
    HIPCHECK(hipHostMalloc(&e->h_graph, sizeof(double) * (W + 1), hipHostMallocDefault));
    HIPCHECK(hipEventCreateWithFlags(&e->ev_graph, sync_event_flags()));
    e->band_cap = N / 2 + 2;
    HIPCHECK(hipMalloc(&e->d_band, sizeof(double) * e->band_cap * FFT_BATCH));
    HIPCHECK(hipMalloc(&e->d_meter, sizeof(double)));
    HIPCHECK(hipMemset(e->d_meter, 0, sizeof(double)));
    HIPCHECK(hipMalloc(&e->d_graph, sizeof(double) * (W + 1)));      // [W] = the S-meter sum at release
    HIPCHECK(hipMalloc(&e->d_pix, sizeof(double) * W));
    HIPCHECK(hipMalloc(&e->d_k0, sizeof(int) * W));
    return QK_OK;
}

// `frames` complete, consecutive frames of fft_size samples at `frame` (device): window, FFT, S-meter bins, magnitude average.
// Only the four-step transform takes more than one frame per launch (frames <= FFT_BATCH).
static void fft_frames(qk_engine *e, const double2 *frame, int frames)
{
    const int N = e->fftN;
    FftArgs a{};
    a.x = frame; a.window = e->d_window; a.tw4096 = e->d_tw4096; a.tw_lo = e->d_tw_lo; a.tw_hi = e->d_tw_hi;
    a.Y = e->d_fft_Y; a.avg = e->d_fft_avg; a.band = e->d_band; a.N = N; a.N1 = e->fftN1;
    a.frames = frames; a.band_stride = e->band_cap; a.acc_parts = e->fft_parts;
    // S-meter pass band of receiver 0 (quisk.c:4659-4667)
    const int fs = e->cfg.sample_rate;
    const int bw = e->nrx > 0 ? e->bandwidth[0] : 0;
    const int tune = e->nrx > 0 ? e->tune[0] : 0, off = e->nrx > 0 ? e->start_offset[0] : 0;
    const double d2 = (double)bw * N / fs;
    const bool scan = e->scan_blocks > 0;
    const int block = scan ? e->scan_next : 0;
    if (scan) e->scan_next = block + 1 < e->scan_blocks ? block + 1 : 0;          // quisk.c:3680-3689
    int i0 = (int)((double)(tune + off) * N / fs + 0.5);
    if (scan) {                                                                    // quisk.c:4660-4663
        const double d1 = ((double)e->scan_tx_tune + e->scan_vfo_screen - e->scan_vfo0 - e->scan_deltaf * block) * N / e->scan_rate;
        i0 = (int)(d1 - d2 / 2 + 0.5);
    }
    if (scan) {     // the average buffers hold this one frame's magnitudes, which k_scan_stitch then places
        a.acc_parts = 0;
        if (e->fftN1 <= 1) hipMemsetAsync(e->d_fft_avg, 0, sizeof(double) * N, e->fe_stream);
    }
    const int nw = (int)(floor(d2) + 0.01);
    // (a zero bandwidth adds nothing to the sum: no launch)
    const bool meter_on = bw > 0 && (i0 > -N / 2 && i0 + nw + 1 < N / 2) && nw + 1 <= e->band_cap;
    a.band_start = i0; a.band_count = meter_on ? nw + 1 : 0;
    Timer tm(e, QK_KT_FFT, e->fe_stream);
    if (e->blue) {
        blue_run(e, e->blue, e->fe_stream, frame, nullptr, e->d_fft_Y);
        hipLaunchKernelGGL(k_graph_accum, dim3((N + 255) / 256), dim3(256), 0, e->fe_stream, e->d_fft_Y, a);
    } else if (e->fftN1 > 1) {
        const dim3 gc(e->fftN1, frames);      // N1 column tiles of 4096 / N1 columns each
        switch (e->fftN1) {
#define QK_COLS(n1) case n1: if (e->fp32) hipLaunchKernelGGL((k_fft_cols<n1, float2>), gc, dim3(FT16_THREADS), 0, e->fe_stream, a); \
                             else hipLaunchKernelGGL((k_fft_cols<n1, double2>), gc, dim3(FT16_THREADS), 0, e->fe_stream, a); break
        QK_COLS(2); QK_COLS(4); QK_COLS(8); QK_COLS(16); QK_COLS(32); QK_COLS(64); QK_COLS(128); QK_COLS(256);
        QK_COLS(512); QK_COLS(1024); QK_COLS(2048); QK_COLS(4096);
#undef QK_COLS
        }
        const int groups = std::min(frames, FFT_GROUPS);
        if (e->fp32) hipLaunchKernelGGL(k_fft_rows<float2>, dim3(e->fftN1, groups), dim3(FT16_THREADS), 0, e->fe_stream, a);
        else hipLaunchKernelGGL(k_fft_rows<double2>, dim3(e->fftN1, groups), dim3(FT16_THREADS), 0, e->fe_stream, a);
        e->fft_parts = std::max(e->fft_parts, groups);
    } else {
        switch (N) {
        case 64: hipLaunchKernelGGL(k_fft_small<64>, dim3(1), dim3(FFT_THREADS), 0, e->fe_stream, a); break;
        case 128: hipLaunchKernelGGL(k_fft_small<128>, dim3(1), dim3(FFT_THREADS), 0, e->fe_stream, a); break;
        case 256: hipLaunchKernelGGL(k_fft_small<256>, dim3(1), dim3(FFT_THREADS), 0, e->fe_stream, a); break;
        case 512: hipLaunchKernelGGL(k_fft_small<512>, dim3(1), dim3(FFT_THREADS), 0, e->fe_stream, a); break;
        case 1024: hipLaunchKernelGGL(k_fft_small<1024>, dim3(1), dim3(FFT_THREADS), 0, e->fe_stream, a); break;
        case 2048: hipLaunchKernelGGL(k_fft_small<2048>, dim3(1), dim3(FFT_THREADS), 0, e->fe_stream, a); break;
        default: hipLaunchKernelGGL(k_fft_small<4096>, dim3(1), dim3(FFT_THREADS), 0, e->fe_stream, a); break;
        }
    }
    if (meter_on)
        hipLaunchKernelGGL(k_meter_accum, dim3(1), dim3(64), 0, e->fe_stream, e->d_band, nw, d2 - nw, e->d_meter, frames, e->band_cap);
    if (scan) {
        const int m0 = (int)(N * ((1.0 - e->scan_valid) / 2.0)), deltam = (int)(N * e->scan_valid / e->scan_blocks);
        ReleaseArgs ra{e->d_fft_avg, nullptr, nullptr, N, e->fftN1, 0, 0, 0.0, 1, nullptr, 0};
        if (deltam > 0)
            hipLaunchKernelGGL(k_scan_stitch, dim3((deltam + 255) / 256), dim3(256), 0, e->fe_stream, ra, e->d_scan_avg, m0, deltam, block, e->scan_blocks);
        e->fft_parts = 0;
        if (block == e->scan_blocks - 1) e->fft_count += 1;                         // quisk.c:4683-4684
        e->stats.fft_frames += frames;
        return;
    }
    e->fft_count += frames;
    e->stats.fft_frames += frames;
}

// FFT staging (quisk.c:2324-2348): consecutive, non-overlapping frames of the PRE-mixer samples.  The reference
// queues at most four frames for the GUI thread and drops the rest (fft_error); here every frame is
// transformed as soon as it is complete (several at once when a block holds several).
static int raw_frame(qk_engine *e, const double2 *frame)
{   // the frame would be queued for the GUI thread; a full queue drops it (fft_error, quisk.c:2335-2345)
    if (e->raw_count >= 4) { e->stats.fft_error++; return QK_OK; }
    const int W = e->fftW, slot = (e->raw_head + e->raw_count) & 3;
    hipLaunchKernelGGL(k_copy_c, dim3((W + 255) / 256), dim3(256), 0, e->fe_stream, frame, e->d_raw_rows + (size_t)slot * W, W);
    e->raw_count++;
    return QK_OK;
}

static int feed_spectrum(qk_engine *e, const double2 *x, int n)
{
    const int N = e->fftN;
    int pos = 0;
    if (e->raw_mode) {
        while (pos < n) {
            const int take = std::min(N - e->fft_fill, n - pos);
            // only the head of a frame is ever read back: keep the first data_width samples
            if (e->fft_fill < e->fftW) {
                const int c = std::min(take, e->fftW - e->fft_fill);
                if (e->small_mode) hipLaunchKernelGGL(k_copy_c, dim3((c + 255) / 256), dim3(256), 0, e->fe_stream, x + pos, e->d_fft_stage + e->fft_fill, c);
                else HIPCHECK(hipMemcpyAsync(e->d_fft_stage + e->fft_fill, x + pos, sizeof(double2) * c, hipMemcpyDeviceToDevice, e->fe_stream));
            }
            e->fft_fill += take; pos += take;
            if (e->fft_fill == N) { int rc = raw_frame(e, e->d_fft_stage); if (rc != QK_OK) return rc; e->fft_fill = 0; }
        }
        return QK_OK;
    }
    while (pos < n) {
        if (e->fft_fill == 0 && n - pos >= N) {         // whole frames straight from the input block
            const int batch = (e->fftN1 > 1 && e->scan_blocks == 0) ? e->fft_batch : 1;
            const int f = std::min(batch, (n - pos) / N);
            fft_frames(e, x + pos, f);
            pos += f * N;
            continue;
        }
        const int take = std::min(N - e->fft_fill, n - pos);
        // (short blocks: the input is pinned host memory and a launch costs the host less than a copy command)
        if (e->small_mode) hipLaunchKernelGGL(k_copy_c, dim3((take + 255) / 256), dim3(256), 0, e->fe_stream, x + pos, e->d_fft_stage + e->fft_fill, take);
        else HIPCHECK(hipMemcpyAsync(e->d_fft_stage + e->fft_fill, x + pos, sizeof(double2) * take, hipMemcpyDeviceToDevice, e->fe_stream));
        e->fft_fill += take; pos += take;
        if (e->fft_fill == N) { fft_frames(e, e->d_fft_stage, 1); e->fft_fill = 0; }
    }
    return QK_OK;
}

// The release is in two halves.  graph_enqueue queues the binning kernel behind the frames of the average (it writes the
// row to the device row and to pinned host memory, and the S-meter sum behind it), starts a new average and records an event;
// graph_collect waits for that event alone.  Frames queued in between belong to the next average and keep the GPU busy
// while the host takes its row: stream order puts the release in front of them.
static int graph_enqueue(qk_engine *e, double zoom, double deltaf)
{
    const int N = e->fftN, W = e->fftW, count = e->fft_count;
    if (count <= 0) return 0;
    int n = (int)(zoom * (double)N / W + 0.5);                      // quisk.c:4715-4717
    if (n < 1) n = 1;
    // first bin of every pixel (quisk.c:4720-4721); uploaded only when zoom or deltaf changed
    static thread_local std::vector<int> k0;
    k0.resize(W);
    bool alias = false;
    for (int i = 0; i < W; i++) {
        k0[i] = (int)(N * (deltaf / e->cfg.sample_rate + zoom * ((double)i / W - 0.5) + 0.5) + 0.1);
        if (i >= 1 && k0[i] < i && k0[i] + n - 1 >= 0) alias = true;   // would read a pixel already written in place
    }
    if (!e->k0_valid || e->k0_zoom != zoom || e->k0_deltaf != deltaf) {
        HIPCHECK(hipStreamSynchronize(e->fe_stream));              // (a release still reading the old table)
        HIPCHECK(hipMemcpy(e->d_k0, k0.data(), sizeof(int) * W, hipMemcpyHostToDevice));
        e->k0_valid = true; e->k0_zoom = zoom; e->k0_deltaf = deltaf;
    }
    const bool scan = e->scan_blocks > 0;
    ReleaseArgs ra{scan ? e->d_scan_avg : e->d_fft_avg, e->d_graph, e->d_k0, N, e->fftN1, W, n,
                   20.0 * (log10((double)count) + log10((double)N) + 31.0 * log10(2.0)),          // quisk.c:4741-4742
                   (e->fftN1 > 1 && !scan) ? std::max(1, e->fft_parts) : 1, e->d_meter, scan ? 1 : 0, e->h_graph};
    if (alias) hipLaunchKernelGGL(k_graph_release_inplace, dim3(1), dim3(256), 0, e->fe_stream, ra, e->d_pix);
    else hipLaunchKernelGGL(k_graph_release, dim3((W + REL_PIX - 1) / REL_PIX), dim3(64 * REL_PIX), 0, e->fe_stream, ra);
    if (scan)
        HIPCHECK(hipMemsetAsync(e->d_scan_avg, 0, sizeof(double) * N, e->fe_stream));             // quisk.c:4750-4751
    else if (e->fftN1 <= 1)      // (the four-step pass B overwrites the average when it starts a new one: no 8 N bytes of zeros)
        HIPCHECK(hipMemsetAsync(e->d_fft_avg, 0, sizeof(double) * N, e->fe_stream));
    HIPCHECK(hipEventRecord(e->ev_graph, e->fe_stream));
    e->rel_pending = true; e->rel_count = count;
    e->fft_parts = 0;
    e->fft_count = 0;
    return count;
}
static int graph_collect(qk_engine *e, double *out, double *smeter)
{
    const int N = e->fftN, W = e->fftW, count = e->rel_count;
    HIPCHECK(hipEventSynchronize(e->ev_graph));
    e->rel_pending = false;
    e->last_row.assign(e->h_graph, e->h_graph + W);                 // current_graph (quisk.c:4747): IsSquelch reads it
    if (out) memcpy(out, e->h_graph, sizeof(double) * W);
    const double meter = e->h_graph[W];
    e->graph_rows++;
    const double scale = 1.0 / 2147483647.0 / N;                    // quisk.c:4728-4737
    double sm = meter * scale * scale / count;
    sm = sm > 0 ? 10.0 * log10(sm) : -160.0;
    sm += 4.25969;
    e->smeter = sm;
    if (smeter) *smeter = sm;
    return count;
}
static int release_graph(qk_engine *e, double zoom, double deltaf, double *out, int data_width, double *smeter)
{
    if (data_width != e->fftW) return fail(QK_EINVAL, "data_width differs from the engine's");
    if (!e->rel_pending) {
        const int rc = graph_enqueue(e, zoom, deltaf);
        if (rc <= 0) return rc;
    }
    return graph_collect(e, out, smeter);
}

// quisk_calc_audio_graph (quisk.c:903-964) on receiver 0's detected LSB audio: frames of data_width samples, scaled to 0 dB =
// 2^31 - 1 for a real signal behind a Hanning window, magnitudes of audio_fft_count frames added in order of frequency; once
// a graph is ready further blocks are ignored until get_audio_graph has taken it.  `delay`: the samples are read this far back
// (d_delay in front of the call while the SSB squelch is on, quisk.c:1831-1835).
static int audio_graph_block(qk_engine *e, const Stream &s, int delay, hipStream_t cs)
{
    const int N = e->cfg.data_width;
    if (N < 2) return QK_OK;
    if (!e->ag_inited) {            // the first call only makes the plan and the window (quisk.c:915-933)
        e->ag_N = N;
        e->ag_frames = std::max(1, 8000 / N / 5);
        std::vector<double> win(N);
        for (int i = 0; i < N; i++) win[i] = 0.50 - 0.50 * cos(2. * M_PI * i / N);
        int rc = any_init(e, e->ag_fft, N, win.data());
        if (rc != QK_OK) return rc;
        HIPCHECK(hipMalloc(&e->d_ag_frame, sizeof(double2) * N));
        HIPCHECK(hipMalloc(&e->d_ag_X, sizeof(double2) * N));
        HIPCHECK(hipMalloc(&e->d_ag_avg, sizeof(double) * N));
        HIPCHECK(hipMemset(e->d_ag_avg, 0, sizeof(double) * N));
        e->ag_inited = true;
        return QK_OK;
    }
    if (e->ag_ready) return QK_OK;
    const double scale = (pow(2, 31) - 1) * (N / 2.0) * e->ag_frames * 0.5;
    int pos = 0;
    while (pos < s.n) {
        const int take = std::min(s.n - pos, N - e->ag_index);
        hipLaunchKernelGGL(k_frame_collect, dim3((take + 255) / 256), dim3(256), 0, cs, s.view(), 1, pos - delay, take, e->d_ag_frame, e->ag_index, scale);
        pos += take; e->ag_index += take;
        if (e->ag_index >= N) {
            e->ag_index = 0;
            any_run(e, e->ag_fft, cs, e->d_ag_frame, nullptr, e->d_ag_X);
            hipLaunchKernelGGL(k_frame_accum, dim3((N + 255) / 256), dim3(256), 0, cs, e->d_ag_X, N, e->d_ag_avg);
            if (++e->ag_count >= e->ag_frames) { e->ag_ready = 1; e->ag_count = 0; }
        }
    }
    return QK_OK;
}

static int run_chain(qk_engine *e, Chain &c, const Stream &input, const int *d_list, int nlist, hipStream_t cs);
// measure_freq (quisk.c:4850-4920) on receiver 0's decimated stream
static int measure_freq_block(qk_engine *e, const Stream &dec)
{
    const int N = 12000;
    hipStream_t cs = e->stream;
    Chain &c = e->mf_chain;
    if (!c.built) {
        ChainBuilder b(c, dec.cap);
        for (int i = 0; i < 3; i++) b.fir(e->f_hbdec, 2, -1);
        if (dec.hist < b.in_hist) return fail(QK_EINVAL, "internal: decimated stream history too small for measure_freq");
        int rc = alloc_chain_streams(e, c, 1);
        if (rc != QK_OK) return rc;
        std::vector<double> win(N);
        for (int i = 0; i < N; i++) win[i] = 0.50 - 0.50 * cos(2. * M_PI * i / (N - 1));     // quisk.c:4873-4874
        rc = any_init(e, e->mf_fft, N, win.data());
        if (rc != QK_OK) return rc;
        HIPCHECK(hipMalloc(&e->d_mf_frame, sizeof(double2) * N));
        HIPCHECK(hipMalloc(&e->d_mf_X, sizeof(double2) * N));
        HIPCHECK(hipMalloc(&e->d_mf_avg, sizeof(double) * N));
        HIPCHECK(hipMemset(e->d_mf_avg, 0, sizeof(double) * N));
        HIPCHECK(hipStreamSynchronize(e->stream));
        c.built = true;
    }
    int rc = run_chain(e, c, dec, nullptr, 1, cs);
    if (rc != QK_OK) return rc;
    const Stream &o = c.streams.back();
    const int srate = e->decim_srate / 8;
    const int take = std::min(o.n, N - e->mf_index);      // what does not fit the frame is dropped (quisk.c:4881-4884)
    if (take > 0) {
        hipLaunchKernelGGL(k_frame_collect, dim3((take + 255) / 256), dim3(256), 0, cs, o.view(), 2, 0, take, e->d_mf_frame, e->mf_index, 1.0);
        e->mf_index += take;
    }
    if (e->mf_index < N) return QK_OK;
    e->mf_index = 0;
    any_run(e, e->mf_fft, cs, e->d_mf_frame, nullptr, e->d_mf_X);
    hipLaunchKernelGGL(k_frame_accum, dim3((N + 255) / 256), dim3(256), 0, cs, e->d_mf_X, N, e->d_mf_avg);
    if (++e->mf_count < e->mf_mode / 2) return QK_OK;
    // time for a calculation (quisk.c:4898-4918): a few times a minute, on the host
    e->mf_count = 0;
    std::vector<double> avg(N);
    HIPCHECK(hipMemcpyAsync(avg.data(), e->d_mf_avg, sizeof(double) * N, hipMemcpyDeviceToHost, cs));
    HIPCHECK(hipMemsetAsync(e->d_mf_avg, 0, sizeof(double) * N, cs));
    HIPCHECK(hipStreamSynchronize(cs));
    double dmax = 1.e-20;
    int ipeak = 0;
    const int center = N / 2 - e->rit_freq * N / srate;
    const int k = 500 * N / srate;
    for (int i = center - k; i <= center + k; i++)
        if (i >= 0 && i < N && avg[i] > dmax) { dmax = avg[i]; ipeak = i; }
    if (ipeak < 1 || ipeak > N - 2) return QK_OK;
    const double c3 = 1.36 * (avg[ipeak + 1] - avg[ipeak - 1]) / (avg[ipeak - 1] + avg[ipeak] + avg[ipeak + 1]);
    double freq = srate * (2 * (ipeak + c3) - N) / 2 / N;
    freq += e->tune[0];
    e->mf_value = freq;
    return QK_OK;
}

// QK_DEBUG_SKIP (a measuring aid beside QK_TIMELINE; the audio is wrong with it), bits: 1 = no AGC launches, 2 = no chain
// launches at all, 4 = no FIR decimator launches, 8 = no receive-filter launches, 16 = no interpolator launches
static int debug_skip()
{
    static const int v = [] { const char *t = getenv("QK_DEBUG_SKIP"); return t ? atoi(t) : 0; }();
    return v;
}

// run one chain over `nlist` receivers; `input` is the stream feeding stage.in == -1
// May this run of a chain go out as one launch (k_chain_fused)?  Short blocks only (one workgroup per receiver does all the
// work), chains made of the stage kinds the kernel knows, and none of the audio hooks that are kernels of their own.
static bool chain_fusable(const qk_engine *e, const Chain &c, const Stream &input, int nlist)
{
    // (one workgroup per receiver: worth it only while the whole launch is a few thousand samples -- with 128 receivers of
    // 819 samples each the separate, wider kernels are twice as fast)
    if (!e->fuse_chains || (&c == &e->front && !e->fz_defer) || input.n <= 0 || (long long)input.n * std::max(1, nlist) > e->fuse_max_in) return false;
    if ((int)c.stages.size() + 2 > QK_FUSED_MAX) return false;
    for (const auto &sx : c.streams) if (sx.f32) return false;           // (the fused stages read and write doubles)
    for (const auto &st : c.stages) {
        if (st.kind == ST_RESAMP || st.in_after || st.out_after) return false;
        if (st.kind == ST_FIR && (e->fp32 && st.hf ? false : st.h == nullptr)) return false;
        if (st.hook) {
            if (e->any_notch || (st.hook == 1 && e->ssb_sq_enabled)) return false;
            if (st.hook == 1 && &c == &e->cls_chain[CL_SSB] && !c.members.empty() && c.members[0] == 0 && e->mode[0] == QK_LSB &&
                !(e->ag_inited && e->ag_ready)) return false;                     // the audio graph is still collecting frames
        }
    }
    return true;
}

static int run_chain(qk_engine *e, Chain &c, const Stream &input, const int *d_list, int nlist, hipStream_t cs)
{
    const bool fuse = chain_fusable(e, c, input, nlist);
    FusedArgs fz{};
    Chain::FracSlot *frac_slot_used = nullptr;
    double ma_close = 0;
    auto fz_add = [&](int kind, const StreamView &in_v, const StreamView &out_v, int planes, int n_in, int n_out) -> FStage & {
        FStage &f = fz.st[fz.nst++];
        f.kind = kind; f.in = in_v; f.out = out_v; f.planes = planes; f.n_in = n_in; f.n_out = n_out;
        return f;
    };
    for (auto &st : c.stages) {
        const Stream &in = st.in < 0 ? input : c.streams[st.in];
        Stream &out = c.streams[st.out];
        const int n_in = in.n;
        StreamView in_v = in.view(), out_v = out.view();
        if (st.in_after) in_v.hist += c.shared_n1;
        if (st.out_after) out_v.hist += c.shared_n1;
        int &phase = st.phase_share >= 0 ? c.stages[st.phase_share].phase : st.phase;
        switch (st.kind) {
        case ST_FIR: {
            const int n_out = (phase + n_in) / st.D;
            const int first = st.D - 1 - phase;
            phase = (phase + n_in) % st.D;
            out.n = n_out;
            if (n_out > 0 && fuse) {
                FStage &f = fz_add(FS_FIR, in_v, out_v, in.planes, n_in, n_out);
                f.h = (e->fp32 && st.hf) ? (const void *)st.hf : (const void *)st.h; f.sub = (e->fp32 && st.hf) ? 1 : 0;
                f.D = st.D; f.TQP = st.TQP; f.first = first;
            } else if (n_out > 0 && !(debug_skip() & 4)) {
                const bool f32 = e->fp32 && st.hf;
                FirTArgs a{in_v, out_v, f32 ? (const void *)st.hf : (const void *)st.h, st.TQP, first, n_out, d_list, in.f32 ? 1 : 0, out.f32 ? 1 : 0};
                Timer tm(e, st.timer, cs);
                // outputs per lane: 5 (an LDS load feeds 3.3 FMAs), 3 for the /5 stage: its five-branch tile left room for five
                // workgroups per CU only (10 % faster).  /2 alone gains 7 % from the smaller tile too, but its extra workgroups
                // take that and more from the front end they run beside (headline 0.374 -> 0.389 ms), and /3 does not care
                // (QK_FIR_R3 = mask over D overrides, for experiments)
                static const int r3mask = [] { const char *v = getenv("QK_FIR_R3"); return v ? atoi(v) : (1 << 5); }();    // bit D: 3 outputs per lane
                const int rr = (r3mask >> st.D & 1) ? 3 : FT_R;
                const dim3 grid((n_out + FT_NTH * rr - 1) / (FT_NTH * rr), in.planes, nlist);
                const size_t lds = (f32 ? sizeof(float) : sizeof(double)) * st.D * (FT_NTH * rr + st.TQP);
#define QK_FIRR(DD, RR) if (f32) hipLaunchKernelGGL((k_fir_tiled<DD, FT_NTH, RR, float>), grid, dim3(FT_NTH), lds, cs, a); \
                        else hipLaunchKernelGGL((k_fir_tiled<DD, FT_NTH, RR, double>), grid, dim3(FT_NTH), lds, cs, a)
#define QK_FIR(DD) case DD: if (rr == 3) { QK_FIRR(DD, 3); } else { QK_FIRR(DD, FT_R); } break
                switch (st.D) {
                QK_FIR(1); QK_FIR(2); QK_FIR(3); QK_FIR(5);
                default: return fail(QK_EINVAL, "internal: FIR decimation factor without a kernel");
                }
#undef QK_FIRR
#undef QK_FIR
            }
            break;
        }
        case ST_INTERP: {
            out.n = n_in * st.L;
            if (out.n > 0 && fuse) {
                FStage &f = fz_add(FS_INTERP, in.view(), out.view(), 1, n_in, out.n);
                f.h = st.h; f.T = st.T; f.L = st.L; f.delay = 0;
            } else if (out.n > 0 && !(debug_skip() & 16)) {
                InterpArgs a{in.view(), out.view(), st.h, st.T, st.L, n_in, d_list, (st.delayed && e->ssb_sq_enabled) ? SSQ_N : 0};
                Timer tm(e, st.timer, cs);
                hipLaunchKernelGGL(k_interp, grid1(out.n, 1, nlist), dim3(256), 0, cs, a);
            }
            break;
        }
        case ST_HBINT: {
            out.n = n_in * 2;
            if (n_in > 0 && fuse) {
                fz_add(FS_HBINT, in.view(), out.view(), in.planes, n_in, out.n).delay = 0;
            } else if (n_in > 0 && !(debug_skip() & 16)) {
                HbInterpArgs a{in.view(), out.view(), n_in, d_list, (st.delayed && e->ssb_sq_enabled) ? SSQ_N : 0};
                Timer tm(e, st.timer, cs);
                hipLaunchKernelGGL(k_hb_interp, grid1(n_in, in.planes, nlist), dim3(256), 0, cs, a);
            }
            break;
        }
        case ST_RESAMP: {
            const long long span = (long long)n_in * st.L;
            const int n_out = span > phase ? (int)((span - phase + st.D - 1) / st.D) : 0;
            const int idx0 = phase;
            phase = (int)(phase + (long long)n_out * st.D - span);
            out.n = n_out;
            if (n_out > 0) {
                ResampArgs a{in_v, out_v, st.h, st.T, st.L, st.D, idx0, n_out, d_list};
                Timer tm(e, st.timer, cs);
                hipLaunchKernelGGL(k_resample, grid1(n_out, 2, nlist), dim3(256), 0, cs, a);
            }
            break;
        }
        case ST_RXF: {
            out.n = n_in;
            st.total_in += n_in;
            if (n_in > 0 && fuse) {
                fz_add(FS_RXF, in.view(), out.view(), 2, n_in, n_in).sub = st.rxf_kind;
            } else if (n_in > 0 && !(debug_skip() & 8)) {
                // measure_audio's sum rides in the kernel's epilogue where this stage is the one it measures
                const bool ma_here = st.measure_div > 0 && d_list && !c.members.empty() && c.members[0] == 0;
                RxfArgs a{in.view(), out.view(), e->fp32 ? (const void *)e->d_taps_f : (const void *)e->d_taps, e->d_ntaps, e->d_mode, e->tap_cap, n_in,
                          st.rxf_kind, d_list, ma_here ? e->d_ma : nullptr};
                Timer tm(e, st.timer, cs);
                int nmax = 0;
                for (int r : c.members) nmax = std::max(nmax, e->ntaps[r]);
                const int tqp = std::max(FT_QB, (nmax + FT_QB - 1) / FT_QB * FT_QB);
                const dim3 grid((n_in + FT_NTH * FT_R - 1) / (FT_NTH * FT_R), 1, nlist);
                if (e->fp32) hipLaunchKernelGGL((k_rxf_tiled<FT_NTH, FT_R, float>), grid, dim3(FT_NTH), sizeof(float) * (FT_NTH * FT_R + tqp), cs, a);
                else hipLaunchKernelGGL((k_rxf_tiled<FT_NTH, FT_R, double>), grid, dim3(FT_NTH), sizeof(double) * (FT_NTH * FT_R + tqp), cs, a);
            }
            break;
        }
        case ST_FMDISC: {
            out.n = n_in;
            if (n_in > 0 && fuse) {
                fz_add(FS_FMDISC, in.view(), out.view(), 2, n_in, n_in).fs = st.fs;
            } else if (n_in > 0) {
                const int np = (n_in + 255) / 256;      // = grid1's x extent
                FmDiscArgs a{in.view(), out.view(), n_in, st.fs, d_list, e->d_sq_partial, e->sq_pstride};
                Timer tm(e, st.timer, cs);
                hipLaunchKernelGGL(k_fm_disc, grid1(n_in, 1, nlist), dim3(256), 0, cs, a);
                // the FM squelch measures the filtered samples (quisk.c:1892, 1911-1920)
                SquelchArgs q{e->d_sq_partial, e->sq_pstride, np, n_in, d_list, e->d_sq_state, e->d_sq_flag + (e->pipe_blocks & 1) * e->nrx,
                              e->squelch_level};
                hipLaunchKernelGGL(k_fm_squelch, dim3(nlist), dim3(256), 0, cs, q);
            }
            break;
        }
        case ST_IIR: {
            out.n = n_in;
            if (n_in > 0 && fuse) {
                FStage &f = fz_add(FS_IIR, in.view(), out.view(), 1, n_in, n_in);
                f.sub = st.iir_kind; f.a0 = st.a0; f.a1 = st.a1; f.b1 = st.b1; f.iir_state = st.iir_state;
            } else if (n_in > 0) {
                IirArgs a{in.view(), out.view(), n_in, st.iir_kind, st.a0, st.a1, st.b1, st.iir_state, d_list, nlist};
                Timer tm(e, st.timer, cs);
                hipLaunchKernelGGL(k_serial_iir, dim3(nlist), dim3(64), 0, cs, a);
            }
            break;
        }
        case ST_FRAC: {
            // The reference's position recurrence (quisk.c:581-609) does not depend on the data: the host runs it
            // (the same additions in the same order) and hands the kernel, per output, the index of the newest
            // input sample and the interpolation position.
            const double fdecim = e->decim_srate / 48000.0;
            const double step = fdecim - 1;
            Chain::FracSlot &fs = c.frac_slot[c.frac_next];
            c.frac_next = (c.frac_next + 1) % QK_FRAC_SLOTS;
            if (fs.used) HIPCHECK(hipEventSynchronize(fs.ev));       // eight blocks back: only a host far ahead of the GPU waits here
            int n_out = 0;
            for (int i = 0; i < n_in; i++) {
                if (c.frac_dindex < 2) {
                    if (n_out >= c.frac_cap) return fail(QK_ERANGE, "internal: fractional decimator output exceeds its buffer");
                    fs.idx[n_out] = i; fs.pos[n_out] = c.frac_dindex; n_out++;
                    c.frac_dindex += step;
                } else c.frac_dindex -= 1;
            }
            out.n = n_out;
            if (n_out > 0 && fuse) {
                // the fused kernel reads the positions where the host wrote them; the slot's event is recorded behind it
                fz_add(FS_FRAC, in.view(), out.view(), out.planes, n_in, n_out);
                fz.frac_idx = fs.idx; fz.frac_pos = fs.pos;
                frac_slot_used = &fs;
            } else if (n_out > 0) {
                hipLaunchKernelGGL(k_stage_frac, dim3((n_out + 255) / 256), dim3(256), 0, cs, fs.idx, fs.pos,
                                   c.d_frac_idx, c.d_frac_pos, n_out);
                HIPCHECK(hipEventRecord(fs.ev, cs));
                fs.used = true;
            }
            if (n_out > 0 && !fuse) {
                FracArgs a{in.view(), out.view(), c.d_frac_idx, c.d_frac_pos, n_out, d_list};
                Timer tm(e, st.timer, cs);
                hipLaunchKernelGGL(k_fracdecim, grid1(n_out, out.planes, nlist), dim3(256), 0, cs, a);
            }
            break;
        }
        }
        if (st.out == c.shared && !st.out_after) c.shared_n1 = out.n;     // the first user's samples of this block are in
        if (st.measure_div > 0 && out.n > 0 && d_list && !c.members.empty() && c.members[0] == 0) {
            // measure_audio: bank 0 only (quisk.c:1787-1790, 1873-1876, 1901-1904, 1980-1985, 2015-2018)
            if (fuse) fz_add(FS_MEASURE, out.view(), out.view(), out.planes, out.n, out.n);
            else if (st.kind != ST_RXF) hipLaunchKernelGGL(k_measure_sum, dim3(out.n > 4096 ? std::min(32, (out.n + 1023) / 1024) : 1), dim3(256), 0, cs, out.view(), out.planes, out.n, e->d_ma);
            e->ma_count += out.n;
            const double fsr = (double)(e->decim_srate / st.measure_div);
            if ((double)e->ma_count >= fsr * e->ma_time) {
                ma_close = (double)e->ma_count;        // (behind the fused launch, if there is one)
                if (!fuse) hipLaunchKernelGGL(k_measure_close, dim3(1), dim3(1), 0, cs, e->d_ma, ma_close);
                e->ma_count = 0;
            }
        }
        if (st.hook && out.n > 0 && !fuse) {
            const int rate = e->decim_srate / st.hook_div;
            const bool cw = &c == &e->cls_chain[CL_CW];
            if (e->any_notch) {
                NotchArgs na{out.view(), out.n, rate, cw ? e->rit_freq : 0, d_list, e->d_notch_on, e->d_mode, e->d_notch, e->d_tw4096};
                hipLaunchKernelGGL(k_auto_notch, dim3(nlist), dim3(256), 0, cs, na);
            }
            if (st.hook == 1 && e->ssb_sq_enabled) {
                SsbSquelchArgs sa{out.view(), out.n, rate, e->ssb_sq_level, d_list, e->d_bw, e->d_ssb_carry, e->d_ssb_state,
                                  e->d_sq_flag + (e->pipe_blocks & 1) * e->nrx, e->d_tw4096};
                hipLaunchKernelGGL(k_ssb_squelch, dim3(nlist), dim3(256), 0, cs, sa);
            }
            if (st.hook == 1 && &c == &e->cls_chain[CL_SSB] && !c.members.empty() && c.members[0] == 0 && e->mode[0] == QK_LSB) {
                int rca = audio_graph_block(e, out, e->ssb_sq_enabled ? SSQ_N : 0, cs);     // LSB only (quisk.c:1835)
                if (rca != QK_OK) return rca;
            }
        }
    }
    if (c.shared >= 0) c.streams[c.shared].n += c.shared_n1;      // what the ring took in this block: both users' samples
    if (fuse && &c == &e->front) {          // deferred: the class chain's launch takes these stages along
        e->fz_front = fz;
        return QK_OK;
    }
    if (e->fz_front.nst > 0 && &c != &e->front) {
        if (fuse && fz.nst > 0 && fz.nst + e->fz_front.nst <= QK_FUSED_MAX && nlist == e->nrx) {
            const int k = e->fz_front.nst;
            for (int i = fz.nst - 1; i >= 0; i--) fz.st[i + k] = fz.st[i];
            for (int i = 0; i < k; i++) fz.st[i] = e->fz_front.st[i];
            fz.nst += k;
        } else {                            // (the class chain did not fuse after all: the waiting stages go out alone)
            FusedArgs &ff = e->fz_front;
            ff.rx_list = nullptr; ff.n_rx = e->nrx; ff.tap_cap = e->tap_cap;
            if (e->fp32) hipLaunchKernelGGL(k_chain_fused<float>, dim3(e->nrx), dim3(256), 0, cs, ff);
            else hipLaunchKernelGGL(k_chain_fused<double>, dim3(e->nrx), dim3(256), 0, cs, ff);
        }
        e->fz_front.nst = 0;
    }
    if (fuse && fz.nst > 0) {
        fz.rx_list = d_list; fz.n_rx = nlist; fz.tap_cap = e->tap_cap;
        fz.taps = e->fp32 ? (const void *)e->d_taps_f : (const void *)e->d_taps; fz.ntaps = e->d_ntaps; fz.mode = e->d_mode;
        fz.sq_state = e->d_sq_state; fz.sq_flag = e->d_sq_flag + (e->pipe_blocks & 1) * e->nrx; fz.sq_level = e->squelch_level;
        fz.ma_acc = e->d_ma;
        {
            Timer tm(e, QK_KT_DEMOD, cs);
            if (e->fp32) hipLaunchKernelGGL(k_chain_fused<float>, dim3(nlist), dim3(256), 0, cs, fz);
            else hipLaunchKernelGGL(k_chain_fused<double>, dim3(nlist), dim3(256), 0, cs, fz);
        }
        if (frac_slot_used) { HIPCHECK(hipEventRecord(frac_slot_used->ev, cs)); frac_slot_used->used = true; }
        if (ma_close > 0) hipLaunchKernelGGL(k_measure_close, dim3(1), dim3(1), 0, cs, e->d_ma, ma_close);
    }
    return QK_OK;
}

static void add_tails(TailArgs &ta, Chain &c, int skip, const int *rx_list = nullptr, int n_rx = 0)
{
    for (int i = 0; i < (int)c.streams.size(); i++) {
        Stream &s = c.streams[i];
        if (i == skip || s.hist <= 0 || s.n <= 0) continue;
        if (ta.count >= QK_MAX_TAILS) break;
        ta.t[ta.count++] = TailDesc{s.d, s.rx_stride(), s.hist + s.cap, s.planes, s.hist, s.n, s.d, s.f32 ? 1 : 0, rx_list, n_rx};
    }
}

template <int S>
static void launch_frontend(qk_engine *e, const FrontArgs &fa, int nspans)
{
    // a fixed number of workgroups per receiver walks the spans: with fe_share < 100 the front end leaves part of
    // every CU's LDS and registers to the kernels of the other two streams
    int gx = nspans;
    if (e->fe_share < 100) gx = std::max(1, std::min(nspans, (int)((long long)e->fe_slots * e->fe_share / 100 / std::max(1, e->nrx))));
    FrontArgs fb = fa;
    if (gx <= 2 && (fa.hist_out || fa.head_src)) { fb.helpers = 2; gx += 2; }      // (a span walk of one or two workgroups: the copies get their own)
    if (e->fp32 && e->fe_packed) hipLaunchKernelGGL((k_frontend_pk<S>), dim3(gx, e->nrx), dim3(64), 0, e->fe_stream, fb);
    else if (e->fp32) hipLaunchKernelGGL((k_frontend_wave<S, float>), dim3(gx, e->nrx), dim3(64), 0, e->fe_stream, fb);
    else hipLaunchKernelGGL((k_frontend_wave<S, double>), dim3(gx, e->nrx), dim3(64), 0, e->fe_stream, fb);
}

// NoiseBlanker (quisk.c:2322): five launches on the front end's stream
static const double2 *noise_blanker(qk_engine *e, const double2 *x, int n)
{
    hipStream_t st = e->fe_stream;
    const int s = e->nb_set, nseg = (n + NB_SEG - 1) / NB_SEG;
    NbArgs a{};
    a.x = x; a.n = n;
    a.ext_x = e->d_nb_x[s]; a.ext_mag = e->d_nb_mag[s]; a.ext_p = e->d_nb_p[s];
    a.H = e->nb_H; a.hw = e->nb_hw; a.save = 3 * e->nb_hw;
    a.limit = e->nb_level == 2 ? 4.0 : e->nb_level == 3 ? 2.5 : 6.0;            // quisk.c:652-663
    a.loc = e->d_nb_loc; a.segsum = e->d_nb_segsum; a.segbase = e->d_nb_segbase; a.S = e->d_nb_S;
    a.words = e->d_nb_words; a.any = e->d_nb_any; a.y = e->d_nb_y;
    hipLaunchKernelGGL(k_nb_mag, dim3(nseg), dim3(64), 0, st, a);
    hipLaunchKernelGGL(k_nb_chain, dim3(1), dim3(64), 0, st, a, nseg);
    hipLaunchKernelGGL(k_nb_flags, dim3((n + 255) / 256), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_nb_words, dim3((a.H + n + NB_SEG - 1) / NB_SEG), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_nb_apply, dim3((n + 255) / 256), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_nb_carry, dim3((a.H + 255) / 256), dim3(256), 0, st, a, e->d_nb_x[s ^ 1], e->d_nb_mag[s ^ 1], e->d_nb_p[s ^ 1]);
    e->nb_set = s ^ 1;
    return e->d_nb_y;
}

// DCremove, spectrum inversion and the noise blanker, in the reference's order, in front of everything else; returns the
// buffer the rest of the block reads
static const double2 *dc_and_inversion(qk_engine *e, const double2 *x, int n)
{
    if (e->dc_bw <= 0 && !e->invert) return x;
    hipStream_t st = e->fe_stream;
    const int fs = e->cfg.sample_rate;
    if (e->dc_bw > 1) {
        if (e->dc_alpha_bw != e->dc_bw) {                 // sound.c:179-191
            const double omega = M_PI * e->dc_bw / (fs / 2.0);
            const double Qsin = sin(omega), Qcos = cos(omega), H0 = 1.0 / sqrt(2.0);
            double xx = ((Qcos - 1) * (Qcos - 1) + Qsin * Qsin) / (H0 * H0) - Qsin * Qsin;
            xx = sqrt(xx);
            e->dc_alpha = Qcos - xx;
            e->dc_alpha_bw = e->dc_bw;
        }
        const int nseg = (n + DC_SEG - 1) / DC_SEG;
        DcArgs a{x, e->d_cond, n, e->dc_alpha, pow(e->dc_alpha, (double)DC_SEG), e->d_dc_seg_end, e->d_dc_seg_in, e->invert};
        hipLaunchKernelGGL(k_dc_iir<0>, dim3(nseg), dim3(64), 0, st, a);
        hipLaunchKernelGGL(k_dc_chain, dim3(1), dim3(64), 0, st, e->d_dc_seg_end, e->d_dc_seg_in, nseg, a.alpha_seg,
                           pow(e->dc_alpha, (double)(n - (nseg - 1) * DC_SEG)), e->d_dc_state);
        hipLaunchKernelGGL(k_dc_iir<1>, dim3(nseg), dim3(64), 0, st, a);
        return e->d_cond;
    }
    if (e->dc_bw == 1) {                                  // sound.c:198-222 with key_state == 0 (receive)
        if (e->dc_key_delay < fs) {
            e->dc_key_delay += n;
        } else {
            e->dc_count += n;
            const int close_stretch = e->dc_count > 2LL * fs;
            hipLaunchKernelGGL(k_dc_partial, dim3(256), dim3(256), 0, st, x, n, e->d_dc_partial);
            hipLaunchKernelGGL(k_dc_final, dim3(1), dim3(64), 0, st, e->d_dc_partial, 256, e->d_dc_state + 1, e->d_dc_state + 2,
                               close_stretch, (double)e->dc_count);
            if (close_stretch) e->dc_count = 0;
        }
        hipLaunchKernelGGL(k_dc_sub, dim3((n + 255) / 256), dim3(256), 0, st, x, e->d_cond, n, e->d_dc_state + 2, e->invert);
        return e->d_cond;
    }
    hipLaunchKernelGGL(k_dc_sub, dim3((n + 255) / 256), dim3(256), 0, st, x, e->d_cond, n, (const double2 *)nullptr, 1);
    return e->d_cond;
}

static const double2 *condition_input(qk_engine *e, const double2 *x, int n)
{
    if (e->tone_freq != 0 && e->d_tone_tab) {
        // DCremove (in the sound loop, in front of quisk_process_samples), then the tone, then the inversion (quisk.c:2311-2320)
        const int inv = e->invert;
        e->invert = 0;
        x = dc_and_inversion(e, x, n);
        e->invert = inv;
        const int nA = e->nA;
        ToneArgs a{};
        a.x = x; a.y = e->d_tone_buf; a.n = n; a.invert = inv;
        const int md = e->nrx > 0 ? e->mode[0] : QK_USB;                       // rxMode
        a.kind = md == QK_AM ? 1 : (md == QK_FM || md == QK_DGT_FM) ? 2 : 0;
        a.vt = MixTables{e->d_tone_tab, e->d_tone_tab + nA, e->d_tone_tab + nA + 64, nullptr, nullptr, nA};
        a.at = MixTables{e->d_tone_tab + nA + 128, e->d_tone_tab + 2 * nA + 128, e->d_tone_tab + 2 * nA + 192, nullptr, nullptr, nA};
        a.st_in = e->d_tone_state[e->tone_pp]; a.st_out = e->d_tone_state[e->tone_pp ^ 1];
        hipLaunchKernelGGL(k_add_tone, dim3((n + 255) / 256), dim3(256), 0, e->fe_stream, a);
        e->tone_pp ^= 1;
        x = e->d_tone_buf;
        if (e->nb_level > 0) x = noise_blanker(e, x, n);
        return x;
    }
    x = dc_and_inversion(e, x, n);
    if (e->nb_level > 0) x = noise_blanker(e, x, n);
    return x;
}

// The key-up envelope (quisk.c:2602-2612), queued behind k_agc_apply (and the routes' k_route_combine).
static int output_stage(qk_engine *e, hipStream_t st)
{
    if (e->keyup_env < 1.0) {
        if (e->keyup_used) HIPCHECK(hipEventSynchronize(e->ev_keyup));      // the previous block's ramp kernel still reads the pinned arrays
        int m_max = 0;
        for (int r = 0; r < e->nrx; r++) { e->h_keyup_counts[r] = e->out_count[r]; m_max = std::max(m_max, e->out_count[r]); }
        const double di = 1. / (e->cfg.playback_rate * 5e-3);               // 5 milliseconds
        int nf = 0;
        double env = e->keyup_env;
        for (int i = 0; i < m_max && nf < e->keyup_cap; i++) {
            env += di;
            if (env > 1.0) { env = 1.0; break; }
            e->h_keyup[nf++] = env;
        }
        e->keyup_env = env;
        if (nf > 0) {
            hipLaunchKernelGGL(k_keyup_ramp, dim3((nf + 255) / 256, e->nrx), dim3(256), 0, st, e->d_out, (long long)e->out_stride, e->h_keyup, nf,
                               e->h_keyup_counts, e->nrx);
            HIPCHECK(hipEventRecord(e->ev_keyup, st));
            e->keyup_used = true;
        }
    }
    return QK_OK;
}

static int route_transfer(qk_engine *e, qk_engine::Route &r, bool to_live);
static int sync_all(qk_engine *e)
{
    HIPCHECK(hipStreamSynchronize(e->fe_stream)); HIPCHECK(hipStreamSynchronize(e->stream));
    HIPCHECK(hipStreamSynchronize(e->agc_stream));
    if (e->apply_stream) HIPCHECK(hipStreamSynchronize(e->apply_stream));
    return QK_OK;
}
static int process_device(qk_engine *e, const double2 *x, int n)
{
    if (n < 0 || n > e->cfg.max_block) return fail(QK_ERANGE, "block larger than max_block");
    if (n == 0) { for (auto &c : e->out_count) c = 0; return QK_OK; }
    Timer block_timer(e, 8);
    x = condition_input(e, x, n);
    // ---- display spectrum: frames of the samples as they arrive, before the mixer (quisk.c:2324-2348) ----
    // (short blocks: the spectrum's frames are enqueued behind the audio path, off the way of qk_pull_audio)
    const bool spectrum_last = e->small_mode && e->nrx > 0;
    if (e->fftN > 0 && !spectrum_last) { int rcs = feed_spectrum(e, x, n); if (rcs != QK_OK) return rcs; }
    if (e->nrx == 0) {
        HIPCHECK(hipGetLastError());
        e->stats.samples_in += n;
        e->stats.blocks += 1;
        return QK_OK;
    }
    for (auto &r : e->routes) {
        // a route that has been mono for long enough: both channels carry the same samples again, the receiver's class takes over
        if (!r.live || r.stereo || r.drain > 0) continue;
        int rct = sync_all(e);
        if (rct == QK_OK) rct = route_transfer(e, r, false);
        if (rct != QK_OK) return rct;
        r.live = false;
        rct = upload_agc_cls(e);
        if (rct != QK_OK) return rct;
    }
    const int S = e->S;
    Chain &F = e->front;
    // ---- front end ----
    FrontArgs fa{};
    fa.x = x; fa.n = n; fa.N0 = e->N0;
    fa.hist = e->d_mixhist[e->pp]; fa.H = e->H;
    fa.mt = MixTables{e->d_tabA, e->d_tabB, e->d_tabC, e->d_v0[e->pp], e->d_mix_on, e->nA};
    fa.J0 = e->N0 >> S;
    fa.g_lo0 = fa.J0 << S;
    fa.g_end = ((e->N0 + n) >> S) << S;
    const long long consumed = fa.g_end - fa.g_lo0;
    F.streams[0].n = (int)(consumed >> S);
    // The front end's output stream is double-buffered: this block goes into the twin, which the chains of the
    // block before last were the last to read (ev_read of that twin).
    const bool fe_writes = F.streams[0].n > 0;
    const int q = e->fe_blocks & 1;
    if (fe_writes) {
        std::swap(F.streams[0].d, F.front_alt);
        if (e->read_pending[q]) HIPCHECK(hipStreamWaitEvent(e->fe_stream, e->ev_read[q], 0));
    }
    fa.out = F.streams[0].view();
    fa.out_f32 = F.streams[0].f32 ? 1 : 0;
    tl_mark(e, 0, e->fe_stream);
    {
        Timer tm(e, QK_KT_FRONTEND, e->fe_stream);
        if (S == 0) {
            hipLaunchKernelGGL(k_mix_only, dim3((n + 255) / 256, e->nrx), dim3(256), 0, e->fe_stream, fa);
        } else if (consumed > 0) {
            // spans: enough workgroups to fill 256 CUs several times over, but long enough that the warm-up
            // (44 * 2^S samples per span) stays a small fraction
            const long long unit = (e->fp32 && e->fe_packed) ? PK_CHUNK : FW_CHUNK;
            const long long target_wgs = (long long)e->fe_rounds * e->fe_slots;
            long long want = std::max(1LL, target_wgs / std::max(1, e->nrx));
            long long span = (consumed + want - 1) / want;
            span = std::max(span, 4 * unit);
            static const int span_env = [] { const char *v = getenv("QK_FE_SPAN"); return v ? atoi(v) : 0; }();      // (experiments)
            if (span_env > 0) span = span_env;
            span = (span + unit - 1) / unit * unit;
            const int nspans = (int)((consumed + span - 1) / span);
            fa.span_len = (int)span; fa.nspans = nspans;
            fa.hist_out = e->d_mixhist[e->pp ^ 1]; fa.v0_next = e->d_v0[e->pp ^ 1];
            if (fe_writes && F.streams[0].hist > 0 && F.front_last_n > 0) {     // the head copy rides along too
                fa.head_src = F.front_alt; fa.head_hist = F.streams[0].hist; fa.head_n = F.front_last_n;
            }
            switch (S) {
            case 1: launch_frontend<1>(e, fa, nspans); break;
            case 2: launch_frontend<2>(e, fa, nspans); break;
            case 3: launch_frontend<3>(e, fa, nspans); break;
            case 4: launch_frontend<4>(e, fa, nspans); break;
            default: launch_frontend<5>(e, fa, nspans); break;
            }
        }
    }
    if (!fa.hist_out) {
        // (no front-end launch in this block, or the mixer alone: the history and the tune vector move on by a launch of their own)
        hipLaunchKernelGGL(k_mixhist_update, dim3(S > 0 ? (e->H + 255) / 256 : 1, e->nrx), dim3(256), 0, e->fe_stream, fa,
                           e->d_mixhist[e->pp ^ 1], e->d_v0[e->pp ^ 1]);
    }
    e->pp ^= 1;
    e->N0 += n;
    if (fe_writes) {
        // head copy: the last `hist` samples of the previous block (in the twin) go in front of this one.  On the main stream,
        // beside the front end: what it reads the previous block's front end wrote (the main stream waited for it then), what
        // it writes lies in front of the front end's first output and was last read by the chains of the block before last
        Stream &f0 = F.streams[0];
        TailArgs th{}; th.count = 1; th.rx_list = nullptr;
        th.t[0] = TailDesc{F.front_alt, f0.rx_stride(), f0.hist + f0.cap, f0.planes, f0.hist, F.front_last_n, f0.d, f0.f32 ? 1 : 0};
        if (f0.hist > 0 && F.front_last_n > 0 && !fa.head_src)
            hipLaunchKernelGGL(k_save_tails, dim3(1, 2, e->nrx), dim3(256), 0, e->stream, th);
        F.front_last_n = f0.n;
    }
    {
        hipEvent_t evf = (e->pipe_blocks & 1) ? e->ev_fe2 : e->ev_fe;
        if (!e->small_mode) HIPCHECK(hipEventRecord(evf, e->fe_stream));     // (short blocks: one stream, the order is the stream's own)
        tl_mark(e, 1, e->fe_stream);
        HIPCHECK(hipStreamWaitEvent(e->stream, evf, 0));
        tl_mark(e, 2, e->stream);
    }
    // ---- remaining integer decimation, all receivers ----
    // short blocks, one class with every receiver in order, nothing between the front chain and it: one fused launch for both
    e->fz_defer = false; e->fz_front.nst = 0;
    if (e->small_mode && e->fuse_chains && e->mf_mode == 0 && e->routes.empty()) {
        int active = 0, cl_a = -1;
        for (int cl = 0; cl < CL_COUNT; cl++) if (!e->cls_chain[cl].members.empty()) { active++; cl_a = cl; }
        if (active == 1 && (int)e->cls_chain[cl_a].members.size() == e->nrx) {
            bool ident = true;
            for (int i = 0; i < e->nrx; i++) if (e->cls_chain[cl_a].members[i] != i) ident = false;
            e->fz_defer = ident;
        }
    }
    const int dbg_skip = debug_skip();
    int rc = (dbg_skip & 2) ? QK_OK : run_chain(e, F, F.streams[0], nullptr, e->nrx, e->stream);
    if (rc != QK_OK) return rc;
    const Stream &dec = F.streams[F.aud];
    if (e->mf_mode > 0 && dec.n > 0) { rc = measure_freq_block(e, dec); if (rc != QK_OK) return rc; }      // quisk.c:2400-2401
    // ---- per class: demodulate and audio chain.  The classes are independent of each other and each is a string of a
    //      dozen short, launch-latency-bound kernels: with more than one class present they go down two lanes, the main
    //      stream and the apply stream (idle at this point of the block, and a stream of its own would share a hardware
    //      queue with one of the four), so that the strings run side by side. ----
    int lane_of[CL_COUNT] = {0};
    hipStream_t lanes[3] = {e->stream, e->apply_stream, e->agc_stream};
    // (the AGC's stream serves as a third lane where the AGC's own pass is short: few audio samples per block)
    const int n_lanes = !e->apply_stream ? 1 : (e->last_m_max > 0 && e->last_m_max < 2048 ? 3 : 2);
    bool lane_used[3] = {true, false, false};
    {
        int load[3] = {0, 0, 0};
        for (int cl = 0; cl < CL_COUNT; cl++) {
            const Chain &c = e->cls_chain[cl];
            if (c.members.empty()) continue;
            int l = 0;
            for (int k = 1; k < n_lanes; k++) if (load[k] < load[l]) l = k;
            lane_of[cl] = l;
            load[l] += (int)c.stages.size();
            lane_used[l] = true;
        }
    }
    if (lane_used[1] || lane_used[2]) {
        if (!e->small_mode) HIPCHECK(hipEventRecord(e->ev_front, e->stream));
        for (int k = 1; k < 3; k++) if (lane_used[k]) HIPCHECK(hipStreamWaitEvent(lanes[k], e->ev_front, 0));
    }
    for (int cl = 0; cl < CL_COUNT; cl++) {
        Chain &c = e->cls_chain[cl];
        const int nl = (int)c.members.size();
        if (!nl) continue;
        // the play-rate stream is double-buffered: this block writes the twin whose history the previous block's
        // tail copy prepared, while the AGC pass of the previous block may still read the other one
        std::swap(c.streams[c.play].d, c.play_alt);
        rc = (dbg_skip & 2) ? QK_OK : run_chain(e, c, dec, c.d_members, nl, lanes[lane_of[cl]]);
        if (rc != QK_OK) return rc;
        c.n_out = c.streams[c.play].n;
        if (c.n_out == 0) std::swap(c.streams[c.play].d, c.play_alt);     // nothing written: stay on the same twin
        if (c.n_out > (int)e->out_stride) return fail(QK_ERANGE, "internal: audio block exceeds the output stride");
        for (int r : c.members) e->out_count[r] = c.n_out;
    }
    // ---- AGC on its own stream: serial in time per receiver (one wave each), so it runs beside the next
    //      block's front end and FIR chain instead of in front of them ----
    const int par = (int)(e->pipe_blocks & 1);
    for (int k = 1; k < 3; k++) {
        // the other lanes join: the AGC, the tail copies and the next block's decimators (which overwrite what these
        // chains read) all come behind them
        if (!lane_used[k]) continue;
        if (!e->small_mode) HIPCHECK(hipEventRecord(e->ev_lane[k - 1], lanes[k]));
        HIPCHECK(hipStreamWaitEvent(e->stream, e->ev_lane[k - 1], 0));
    }
    // ---- stereo routes: the two detected streams, count-equalised, become one pair that runs its own tail chain ----
    for (auto &r : e->routes) {
        if (!r.live) continue;
        Chain &ca = e->cls_chain[e->cls[r.rx]], &cb = e->cls_chain[e->cls[r.partner]];
        const Stream &sa = ca.streams[ca.aud], &sb = r.stereo ? cb.streams[cb.aud] : sa;
        const int na = sa.n, nb = sb.n;
        if (na + r.nbuf_a >= e->route_cap || nb + r.nbuf_b >= e->route_cap) r.nbuf_a = r.nbuf_b = 0;   // overflow: drop what waits (quisk.c:1465-1469)
        const int ta = r.nbuf_a + na, tb = r.nbuf_b + nb, nout = std::min(ta, tb);
        Chain &t = r.tail;
        std::swap(t.streams[t.play].d, t.play_alt);
        if (nout > t.streams[0].cap) return fail(QK_ERANGE, "internal: a route's block exceeds its pair stream");
        RouteArgs a{sa.view(), sb.view(), r.rx, r.stereo ? r.partner : r.rx, r.stereo ? r.method : 4, r.fifo[r.cur][0], r.fifo[r.cur][1],
                    r.fifo[r.cur ^ 1][0], r.fifo[r.cur ^ 1][1], r.nbuf_a, r.nbuf_b, na, nb, nout, t.streams[0].view()};
        const int span = std::max(ta, tb);
        if (span > 0) hipLaunchKernelGGL(k_route_pair, dim3((span + 255) / 256), dim3(256), 0, e->stream, a);
        r.nbuf_a = ta - nout; r.nbuf_b = tb - nout; r.cur ^= 1;
        t.streams[0].n = nout;
        rc = run_chain(e, t, t.streams[0], nullptr, 2, e->stream);
        if (rc != QK_OK) return rc;
        t.n_out = t.streams[t.play].n;
        if (t.n_out == 0) std::swap(t.streams[t.play].d, t.play_alt);
        if (t.n_out > (int)e->out_stride) return fail(QK_ERANGE, "internal: audio block exceeds the output stride");
        e->out_count[r.rx] = t.n_out;
    }
    if (!e->small_mode) HIPCHECK(hipEventRecord(e->ev_chain, e->stream));
    tl_mark(e, 3, e->stream);
    HIPCHECK(hipStreamWaitEvent(e->agc_stream, e->ev_chain, 0));
    tl_mark(e, 4, e->agc_stream);
    if (fe_writes) {
        if (!e->small_mode) HIPCHECK(hipEventRecord(e->ev_read[q], e->stream));
        e->read_pending[q] = true;
        e->fe_blocks++;
    }
    hipStream_t agc_ev_stream = e->agc_stream;
    {
        static_assert(CL_COUNT <= QK_AGC_CLASSES, "AgcArgs holds one stream view per chain class");
        const int rate = e->cfg.playback_rate;
        AgcArgs a{};
        int m_max = 0;
        bool any_cpx = false;
        for (int cl = 0; cl < CL_COUNT; cl++) {
            Chain &c = e->cls_chain[cl];
            const bool on = !c.members.empty() && c.n_out > 0;
            if (c.built) a.in_c[cl] = c.streams[c.play].view();
            a.n_c[cl] = on ? c.n_out : 0;
            a.cpx_c[cl] = cl == CL_DGTIQ;
            a.mag_c[cl] = a.in_c[cl];
            if (a.cpx_c[cl]) { a.mag_c[cl] = StreamView{e->d_agc_mag, e->gain_stride, e->gain_stride, 0}; any_cpx = any_cpx || on; }
            m_max = std::max(m_max, a.n_c[cl]);
        }
        e->last_m_max = m_max;
        if (m_max > 0 && !(dbg_skip & 1)) {
            a.cls = e->d_agc_cls; a.gains = (e->apply_stream && par) ? e->d_gains2 : e->d_gains; a.gain_stride = e->gain_stride;
            a.buf_size = rate * 15 / 1000; a.n_rx = e->nrx;
            a.ceiling = 0.7 * QK_CLIP32;                                         // Agc1 max_out, quisk.c:2181
            a.time_release = 1.0 - exp(-1.0 / rate / e->agc_release_time);       // quisk.c:2045
            a.release_gain = e->d_release_gain; a.mode = e->d_mode; a.state = e->d_agc;
            a.out = e->d_out; a.out_stride = (long long)e->out_stride;
            a.started = e->d_agc_started; a.abort_flag = e->d_agc_abort;
            a.mute = e->d_sq_flag + par * e->nrx; a.kill = e->kill_audio;
            Timer tm(e, QK_KT_AGC, e->agc_stream);
            if (any_cpx) hipLaunchKernelGGL(k_agc_cabs, grid1(m_max, 1, e->nrx), dim3(256), 0, e->agc_stream, a);
            // (short blocks are bound by the host's launches, and the second stream costs two more calls: it is used from
            // 4096 audio samples per receiver on)
            const bool own = e->apply_stream && m_max >= 4096;
            a.apply_here = (e->small_mode && !own) ? 1 : 0;
            // The gain pass sets the block period at low sample rates and runs back to back on this stream: the event the
            // apply stream waits for rides on the kernel's own completion signal (hipExtLaunchKernelGGL) instead of a
            // marker packet of its own behind it.
            static const bool ext_launch = [] { const char *v = getenv("QK_AGC_EXT_LAUNCH"); return !v || atoi(v) != 0; }();
            if (own && ext_launch) hipExtLaunchKernelGGL(k_agc_gain, dim3(e->nrx), dim3(192), 0, e->agc_stream, nullptr, e->ev_gain, 0, a);
            else hipLaunchKernelGGL(k_agc_gain, dim3(e->nrx), dim3(192), 0, e->agc_stream, a);
            tl_mark(e, 5, e->agc_stream);
            hipStream_t aps = e->agc_stream;
            if (!own && e->last_apply_own && e->agc_pending)
                HIPCHECK(hipStreamWaitEvent(e->agc_stream, e->ev_agc[par ^ 1], 0));      // the previous apply pass, on the other stream
            e->last_apply_own = own;
            agc_ev_stream = own ? e->apply_stream : e->agc_stream;
            if (own) {
                if (!ext_launch) HIPCHECK(hipEventRecord(e->ev_gain, e->agc_stream));
                HIPCHECK(hipStreamWaitEvent(e->apply_stream, e->ev_gain, 0));
                aps = e->apply_stream;
                a.started = e->d_agc_started2;
            }
            if (!a.apply_here) hipLaunchKernelGGL(k_agc_apply, grid1(m_max, 1, e->nrx), dim3(256), 0, aps, a);
            for (int cl = 0; cl < CL_COUNT; cl++) {
                if (a.n_c[cl] <= 0) continue;
                for (int r : e->cls_chain[cl].members)
                    if (!e->agc_started[r]) {
                        hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(e->d_agc_started + r), 1, 1, e->agc_stream);
                        if (e->apply_stream) hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(e->d_agc_started2 + r), 1, 1, aps);
                    }
            }
        }
    }
    for (auto &r : e->routes) {
        // ---- a route's two AGCs: Agc1 on the left channel, Agc2 on the right (quisk.c:2563-2572), each a mono process_agc;
        //      then both channels into the receiver's output row, the squelch per channel (quisk.c:2585-2601) ----
        Chain &t = r.tail;
        if (!r.live || t.n_out <= 0) continue;
        const int rate = e->cfg.playback_rate, m = t.n_out;
        AgcArgs a{};
        a.in_c[0] = t.streams[t.play].view(); a.mag_c[0] = a.in_c[0]; a.n_c[0] = m;
        a.cls = e->d_zero2; a.gains = e->d_gains + (size_t)r.slot * e->gain_stride; a.gain_stride = e->gain_stride;
        a.buf_size = rate * 15 / 1000; a.n_rx = 2;
        a.ceiling = 0.7 * QK_CLIP32; a.time_release = 1.0 - exp(-1.0 / rate / e->agc_release_time);
        a.release_gain = e->d_release_gain + r.slot; a.mode = e->d_mode + r.slot; a.state = e->d_agc + r.slot;
        a.out = e->d_out + (size_t)r.slot * e->out_stride; a.out_stride = (long long)e->out_stride;
        a.started = e->d_agc_started + r.slot; a.abort_flag = e->d_agc_abort;
        a.mute = nullptr; a.kill = 0;
        hipStream_t st = agc_ev_stream;
        if (!r.stereo) {
            hipLaunchKernelGGL(k_agc_gain, dim3(1), dim3(192), 0, st, a);
            DrainArgs da{a.in_c[0], a.gains, a.buf_size, m, a.started, e->d_out + (size_t)r.rx * e->out_stride, e->d_sq_flag + par * e->nrx + r.rx,
                         e->kill_audio};
            hipLaunchKernelGGL(k_route_drain_apply, dim3((m + 255) / 256), dim3(256), 0, st, da);
            if (!e->agc_started[r.slot]) hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(e->d_agc_started + r.slot), 1, 1, st);
            r.drain -= m;
            continue;
        }
        hipLaunchKernelGGL(k_agc_gain, dim3(2), dim3(192), 0, st, a);
        hipLaunchKernelGGL(k_agc_apply, grid1(m, 1, 2), dim3(256), 0, st, a);
        const int src_l = (r.method == 1 || r.method == 4) ? r.rx : r.partner, src_r = (r.method == 2 || r.method == 4) ? r.rx : r.partner;
        const int *fl = e->d_sq_flag + par * e->nrx;
        hipLaunchKernelGGL(k_route_combine, dim3((m + 255) / 256), dim3(256), 0, st, a.out, a.out + e->out_stride,
                           e->d_out + (size_t)r.rx * e->out_stride, m, fl + src_l, fl + src_r, e->kill_audio);
        for (int ch = 0; ch < 2; ch++)
            if (!e->agc_started[r.slot + ch])
                hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(e->d_agc_started + r.slot + ch), 1, 1, st);
    }
    {   // ---- output stage behind the AGC, on the stream that wrote the output rows ----
        int rco = output_stage(e, agc_ev_stream);
        if (rco != QK_OK) return rco;
    }
    tl_mark(e, 6, agc_ev_stream);
    if (!e->small_mode) HIPCHECK(hipEventRecord(e->ev_agc[par], agc_ev_stream));
    else { HIPCHECK(hipEventRecord(e->ev_audio, e->stream)); e->audio_ev = true; }
    // ---- carry stream tails ----
    {
        // (short blocks: the front chain's and the classes' tails leave in ONE launch, each entry with its own receiver list)
        const bool merge = e->small_mode;      // (in the pipelined mode one merged launch measured no shorter: 0.3227 against 0.3217 ms at config 2)
        TailArgs all{}; all.count = 0; all.rx_list = nullptr;
        TailArgs ta{}; ta.count = 0; ta.rx_list = nullptr;
        if (merge) add_tails(all, F, 0, nullptr, e->nrx);
        else add_tails(ta, F, 0);        // stream 0 is carried by the head copy
        if (ta.count) hipLaunchKernelGGL(k_save_tails, dim3(ta.count, 2, e->nrx), dim3(256), 0, e->stream, ta);
        if (e->mf_mode > 0 && e->mf_chain.built) {
            TailArgs tm{}; tm.count = 0; tm.rx_list = nullptr;
            add_tails(tm, e->mf_chain, -1);
            if (tm.count) hipLaunchKernelGGL(k_save_tails, dim3(tm.count, 2, 1), dim3(256), 0, e->stream, tm);
        }
        // the play streams' tails go into the twin buffers, which the previous block's AGC pass may still be reading
        if (e->agc_pending) HIPCHECK(hipStreamWaitEvent(e->stream, e->ev_agc[par ^ 1], 0));
        for (int cl = 0; cl < CL_COUNT; cl++) {
            Chain &c = e->cls_chain[cl];
            const int nl = (int)c.members.size();
            if (!nl) continue;
            TailArgs tb{}; tb.count = 0; tb.rx_list = c.d_members;
            Stream &play = c.streams[c.play];
            int extra = 1;
            for (auto &sx : c.streams) if (sx.hist > 0 && sx.n > 0) extra++;
            const bool into_all = merge && all.count + extra <= QK_MAX_TAILS;
            TailArgs &dst_t = into_all ? all : tb;
            add_tails(dst_t, c, c.play, into_all ? c.d_members : nullptr, into_all ? nl : 0);
            if (play.n > 0 && dst_t.count < QK_MAX_TAILS)
                dst_t.t[dst_t.count++] = TailDesc{play.d, play.rx_stride(), play.hist + play.cap, play.planes, play.hist, play.n, c.play_alt, 0,
                                                  into_all ? c.d_members : nullptr, into_all ? nl : 0};
            if (tb.count) hipLaunchKernelGGL(k_save_tails, dim3(tb.count, 2, nl), dim3(256), 0, e->stream, tb);
            // (the merged launch goes out in front of the first memset below, or behind the loop)
            if (into_all && c.n_out > 0) {
                bool any_first = false;
                for (int r : c.members) if (!e->agc_started[r]) any_first = true;
                if (any_first && all.count) { hipLaunchKernelGGL(k_save_tails, dim3(all.count, 2, e->nrx), dim3(256), 0, e->stream, all); all.count = 0; }
            }
            if (c.n_out > 0) {
                // A receiver's first process_agc call returns the block untouched and does NOT store it in the
                // delay ring (quisk.c:2033-2050): the ring (= the play stream's history) stays zero.
                for (int r : c.members) {
                    if (e->agc_started[r]) continue;
                    for (int pl = 0; pl < play.planes; pl++)
                        hipMemsetAsync(c.play_alt + (long long)r * play.rx_stride() + (long long)pl * (play.hist + play.cap), 0,
                                       sizeof(double) * play.hist, e->stream);
                    e->agc_started[r] = 1;
                }
            }
        }
        if (all.count) hipLaunchKernelGGL(k_save_tails, dim3(all.count, 2, e->nrx), dim3(256), 0, e->stream, all);
        for (auto &r : e->routes) {
            if (!r.live) continue;
            Chain &t = r.tail;
            TailArgs tb{}; tb.count = 0; tb.rx_list = nullptr;
            add_tails(tb, t, t.play);
            Stream &play = t.streams[t.play];
            if (play.n > 0 && tb.count < QK_MAX_TAILS)
                tb.t[tb.count++] = TailDesc{play.d, play.rx_stride(), play.hist + play.cap, play.planes, play.hist, play.n, t.play_alt};
            if (tb.count) hipLaunchKernelGGL(k_save_tails, dim3(tb.count, 2, 2), dim3(256), 0, e->stream, tb);
            if (t.n_out > 0)
                for (int ch = 0; ch < (r.stereo ? 2 : 1); ch++) {        // a first process_agc call does not store its block (quisk.c:2033-2050)
                    if (e->agc_started[r.slot + ch]) continue;
                    hipMemsetAsync(t.play_alt + (long long)ch * play.rx_stride(), 0, sizeof(double) * play.hist, e->stream);
                    if (!r.stereo) hipMemsetAsync(t.play_alt + play.rx_stride(), 0, sizeof(double) * play.hist, e->stream);
                    e->agc_started[r.slot + ch] = 1;
                }
        }
        e->agc_pending = true;
    }
    if (e->fftN > 0 && spectrum_last) { int rcs = feed_spectrum(e, x, n); if (rcs != QK_OK) return rcs; }
    tl_mark(e, 7, e->stream);
    HIPCHECK(hipGetLastError());
    e->stats.samples_in += n;
    e->stats.blocks += 1;
    e->pipe_blocks += 1;
    return QK_OK;
}

// =================================================================================== C ABI
extern "C" {

static int routes_refresh(qk_engine *e);
static int route_sync_params(qk_engine *e);
static int route_stereo(qk_engine *e, qk_engine::Route &r, bool on);

const char *qk_last_error(void) { return g_err.c_str(); }
const char *qk_version(void) { return "quisk_amd 0.1 (gfx950)"; }

int qk_create(qk_engine **out, const qk_config *cfg)
{
    if (!out || !cfg) return fail(QK_EINVAL, "null argument");
    *out = nullptr;
    if (cfg->sample_rate < 41000 || cfg->n_receivers < 0 || cfg->max_block < 1 || (cfg->n_receivers == 0 && cfg->fft_size <= 0))
        return fail(QK_EINVAL, "bad sample_rate / n_receivers / max_block");
    const int pr = cfg->playback_rate / 48000;
    if (cfg->playback_rate % 48000 || (pr != 1 && pr != 2 && pr != 4 && pr != 8))
        return fail(QK_EINVAL, "playback_rate must be 48000 x {1,2,4,8} (quisk.c:2535-2553)");
    if (cfg->reserved[1] != 0 && cfg->reserved[1] != 1) return fail(QK_EINVAL, "qk_config.reserved[1]: 0 = fp64, 1 = fp32 mode");
    if (cfg->reserved[2] < 0 || cfg->reserved[2] > 2) return fail(QK_EINVAL, "qk_config.reserved[2]: short-block mode 0 = by max_block, 1 = on, 2 = off");
    int ndev = 0;       // (the arguments are judged before the device is: the CPU tests reach every message above)
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(QK_EHIP, "no HIP device: the engine has no CPU path");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(QK_EINVAL, "bad device ordinal");
    qk_engine *e = new (std::nothrow) qk_engine();
    if (!e) return fail(QK_ENOMEM, "out of host memory");
    e->cfg = *cfg;
    e->fp32 = cfg->reserved[1] == 1;
    if (cfg->reserved[0] > 0) e->tap_cap = (cfg->reserved[0] + FT_QB - 1) / FT_QB * FT_QB;   // max receive-filter taps
    int rc = engine_init(e);
    if (rc != QK_OK) { std::string keep = g_err; qk_destroy(e); g_err = keep; return rc; }
    *out = e;
    return QK_OK;
}

void qk_destroy(qk_engine *e)
{
    if (!e) return;
    hipSetDevice(e->cfg.device);
    if (e->fe_stream) hipStreamSynchronize(e->fe_stream);
    if (e->stream) hipStreamSynchronize(e->stream);
    if (e->agc_stream) hipStreamSynchronize(e->agc_stream);
    if (e->apply_stream) hipStreamSynchronize(e->apply_stream);
    tl_dump(e);
    free_chain(e->front);
    for (auto &c : e->cls_chain) free_chain(c);
    blue_free(e->blue);
    any_free(e->mrx);
    any_free(e->bs);
    for (void *p : {(void *)e->d_mrx_in, (void *)e->d_mrx_X, (void *)e->d_mrx_out, (void *)e->d_bs_in, (void *)e->d_bs_avg,
                    (void *)e->d_bs_out, (void *)e->d_bs_X, (void *)e->d_bs_max})
        if (p) hipFree(p);
    for (int i = 0; i < 2; i++) { if (e->d_mixhist[i]) hipFree(e->d_mixhist[i]); if (e->d_v0[i]) hipFree(e->d_v0[i]); }
    if (e->small_mode) {
        if (e->in_on_device) { if (e->d_in) hipFree(e->d_in); if (e->d_in_alt) hipFree(e->d_in_alt); }
        else { if (e->d_in) hipHostFree(e->d_in); if (e->d_in_alt) hipHostFree(e->d_in_alt); }
        if (e->d_out) hipHostFree(e->d_out);
        e->d_in = nullptr; e->d_out = nullptr;
    }
    for (auto &ev : e->ev_in) if (ev) hipEventDestroy(ev);
    void *ptrs[] = {e->d_nb_x[0], e->d_nb_x[1], e->d_nb_mag[0], e->d_nb_mag[1], e->d_nb_p[0], e->d_nb_p[1], e->d_nb_y, e->d_nb_loc, e->d_nb_segsum,
                    e->d_nb_segbase, e->d_nb_S, e->d_nb_words, e->d_nb_any, e->d_cond, e->d_dc_seg_end, e->d_dc_seg_in, e->d_dc_state, e->d_dc_partial, e->d_tabA, e->d_tabB, e->d_tabC, e->d_mix_on, e->d_in, e->d_raw, e->d_coefs, e->d_taps, e->d_ntaps, e->d_cls,
                    e->d_mode, e->d_release_gain, e->d_coefs_f, e->d_taps_f, e->d_agc, e->d_agc_started, e->d_gains, e->d_agc_mag, e->d_sq_state, e->d_sq_partial, e->d_sq_flag, e->d_out,
                    e->d_fft_stage, e->d_fft_Y, e->d_tw4096, e->d_tw_lo, e->d_tw_hi, e->d_window, e->d_fft_avg, e->d_band,
                    e->d_meter, e->d_graph, e->d_pix, e->d_k0};
    for (void *p : ptrs) if (p) hipFree(p);
    if (e->h_graph) hipHostFree(e->h_graph);
    if (e->ev_graph) hipEventDestroy(e->ev_graph);
    if (e->h_keyup) hipHostFree(e->h_keyup);
    if (e->h_keyup_counts) hipHostFree(e->h_keyup_counts);
    if (e->ev_keyup) hipEventDestroy(e->ev_keyup);
    if (e->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(e->comm);
    if (e->d_counts) hipFree(e->d_counts);
    for (auto &r : e->routes) { free_chain(r.tail); if (r.d_agc2_ring) hipFree(r.d_agc2_ring); }
    if (e->d_agc_cls) hipFree(e->d_agc_cls);
    if (e->d_zero2) hipFree(e->d_zero2);
    for (void *p : {(void *)e->d_ssb_carry, (void *)e->d_ssb_state, (void *)e->d_bw, (void *)e->d_notch_on, (void *)e->d_notch}) if (p) hipFree(p);
    for (auto &r : e->routes) for (auto &q : r.fifo) for (double *p : q) if (p) hipFree(p);
    for (double *p : e->d_obj) if (p) hipFree(p);
    for (void *p : {(void *)e->d_tone_tab, (void *)e->d_tone_state[0], (void *)e->d_tone_state[1], (void *)e->d_tone_buf,
                    (void *)e->d_gen_in, (void *)e->d_gen_out}) if (p) hipFree(p);
    any_free(e->gen); any_free(e->mf_fft); any_free(e->ag_fft);
    free_chain(e->mf_chain);
    for (void *p : {(void *)e->d_ma, (void *)e->d_mf_frame, (void *)e->d_mf_X, (void *)e->d_mf_avg, (void *)e->d_ag_frame, (void *)e->d_ag_X,
                    (void *)e->d_ag_avg}) if (p) hipFree(p);
    for (void *p : {(void *)e->d_scan_avg, (void *)e->d_raw_rows, (void *)e->d_rfb_x[0], (void *)e->d_rfb_x[1], (void *)e->d_rfb_x[2], (void *)e->d_rfb_y, (void *)e->d_rfb_t}) if (p) hipFree(p);
    for (auto &ev : e->ev_pool) if (ev) hipEventDestroy(ev);
    if (e->agc_stream && e->agc_stream != e->stream) { hipStreamSynchronize(e->agc_stream); hipStreamDestroy(e->agc_stream); }
    if (e->apply_stream && e->apply_stream != e->stream) { hipStreamSynchronize(e->apply_stream); hipStreamDestroy(e->apply_stream); }
    if (e->ev_gain) hipEventDestroy(e->ev_gain);
    if (e->d_gains2) hipFree(e->d_gains2);
    if (e->d_agc_started2) hipFree(e->d_agc_started2);
    if (e->h_agc_abort) hipHostFree(e->h_agc_abort);
    if (e->ev_chain) hipEventDestroy(e->ev_chain);
    if (e->ev_front) hipEventDestroy(e->ev_front);
    for (auto &ev : e->ev_lane) if (ev) hipEventDestroy(ev);
    for (auto &ev : e->ev_agc) if (ev) hipEventDestroy(ev);
    if (e->ev_audio) hipEventDestroy(e->ev_audio);
    if (e->ev_fe) hipEventDestroy(e->ev_fe);
    if (e->ev_fe2) hipEventDestroy(e->ev_fe2);
    for (auto &ev : e->ev_read) if (ev) hipEventDestroy(ev);
    if (e->fe_stream && e->fe_stream != e->stream) hipStreamDestroy(e->fe_stream);
    if (e->stream) hipStreamDestroy(e->stream);
    delete e;
}

#define ENGINE_ENTER(e)                                                  \
    if (!(e)) return fail(QK_EINVAL, "null engine");                     \
    std::lock_guard<std::mutex> _lk((e)->mu);                            \
    HIPCHECK(hipSetDevice((e)->cfg.device))
#define SYNC_ALL(e) do { HIPCHECK(hipStreamSynchronize((e)->stream)); \
                         if (!(e)->small_mode) { HIPCHECK(hipStreamSynchronize((e)->fe_stream)); HIPCHECK(hipStreamSynchronize((e)->agc_stream)); \
                             if ((e)->apply_stream) HIPCHECK(hipStreamSynchronize((e)->apply_stream)); } } while (0)
#define RX_CHECK(e, rx) if ((rx) < 0 || (rx) >= (e)->nrx) return fail(QK_ERANGE, "receiver index out of range")
// after a synchronisation: a hand-over inside k_agc_gain that ran out of polls left NaN gains behind (qk_kernels.h, AGC_SPIN_LIMIT)
#define AGC_ABORT_CHECK(e) do { if ((e)->h_agc_abort && *(e)->h_agc_abort) \
        return fail(QK_EHIP, "AGC kernel: a hand-over between its waves timed out; the audio of this block is not valid"); } while (0)

int qk_set_tune(qk_engine *e, int rx, int tune_hz)
{
    ENGINE_ENTER(e); RX_CHECK(e, rx);
    if (e->tune[rx] == tune_hz) return QK_OK;
    SYNC_ALL(e);
    e->tune[rx] = tune_hz;
    return upload_tune(e, rx);
}

// the tap count the band-pass kernel sees: DGT-IQ is not filtered at 19 kHz and above (quisk.c:1976); N == 0 passes through
static int upload_ntaps(qk_engine *e, int rx)
{
    const int n = (e->mode[rx] == QK_DGT_IQ && e->bandwidth[rx] >= 19000) ? 0 : e->ntaps[rx];
    HIPCHECK(hipMemcpy(e->d_ntaps + rx, &n, sizeof(int), hipMemcpyHostToDevice));
    return QK_OK;
}

int qk_set_mode(qk_engine *e, int rx, int mode)
{
    ENGINE_ENTER(e); RX_CHECK(e, rx);
    const int cl = class_of_mode(mode, e->bandwidth[rx]);
    if (cl < 0) return fail(QK_EUNSUPPORTED, "mode needs an external codec or is not on the GPU path yet");
    if (e->mode[rx] == mode) return QK_OK;
    SYNC_ALL(e);
    e->mode[rx] = mode;
    HIPCHECK(hipMemcpy(e->d_mode + rx, &mode, sizeof(int), hipMemcpyHostToDevice));
    { int rcn = upload_ntaps(e, rx); if (rcn != QK_OK) return rcn; }
    for (int b = 0; b < 2; b++) HIPCHECK(hipMemset(e->d_sq_flag + b * e->nrx + rx, 0, sizeof(int)));     // squelch_active = 0, quisk.c:1777
    if (e->cls[rx] != cl) {
        // (a live route keeps the objects behind the detector -- cFracDecim, HalfBand7..9, the AGCs belong to no mode -- whatever
        // class the receiver is in; only DGT-IQ, stereo by itself, takes them back at once)
        if (cl == CL_DGTIQ)
            for (auto &r : e->routes) {
                if (!r.live || r.rx != rx) continue;
                int rct = r.stereo ? route_stereo(e, r, false) : QK_OK;
                if (rct == QK_OK) rct = route_transfer(e, r, false);
                if (rct != QK_OK) return rct;
                r.live = false; r.drain = 0;
            }
        const int from = e->cls[rx];
        e->cls[rx] = cl;
        int rcm = upload_members(e);
        if (rcm == QK_OK) rcm = migrate_receiver(e, rx, from, cl);
        if (rcm != QK_OK) return rcm;
    }
    return routes_refresh(e);
}

int qk_set_filters(qk_engine *e, int rx, const double *fi, const double *fq, int n, int bandwidth, int start_offset)
{
    ENGINE_ENTER(e); RX_CHECK(e, rx);
    if (n < 0 || (n > 0 && (!fi || !fq))) return fail(QK_EINVAL, "bad filter arguments");
    if (n >= 10001) return fail(QK_ERANGE, "Filter size must be less than 10001");       // quisk.c:4216-4219
    if (n > e->tap_cap)
        return fail(QK_ERANGE, "filter longer than the engine's tap capacity (qk_config.reserved[0])");
    SYNC_ALL(e);
    { int rcr = ring_sync(e, rx, e->cls[rx]); if (rcr != QK_OK) return rcr; }
    {   // the tap order of cRxFilterOut (quisk.c:1176-1186) as a plain FIR: h[0] = f[0], h[m] = f[n - m]; zero-padded
        std::vector<double> hp((size_t)2 * e->tap_cap, 0.0);
        for (int m = 0; m < n; m++) {
            hp[m] = fi[m ? n - m : 0];
            hp[(size_t)e->tap_cap + m] = fq[m ? n - m : 0];
        }
        HIPCHECK(hipMemcpy(e->d_taps + (size_t)rx * 2 * e->tap_cap, hp.data(), sizeof(double) * hp.size(), hipMemcpyHostToDevice));
        if (e->d_taps_f) {
            std::vector<float> hf(hp.begin(), hp.end());
            HIPCHECK(hipMemcpy(e->d_taps_f + (size_t)rx * 2 * e->tap_cap, hf.data(), sizeof(float) * hf.size(), hipMemcpyHostToDevice));
        }
    }
    if (rx == 0) { e->f0_i.assign(fi, fi + n); e->f0_q.assign(fq, fq + n); }
    e->ntaps[rx] = n; e->bandwidth[rx] = bandwidth; e->start_offset[rx] = start_offset;
    HIPCHECK(hipMemcpy(e->d_bw + rx, &bandwidth, sizeof(int), hipMemcpyHostToDevice));
    { int rcn = upload_ntaps(e, rx); if (rcn != QK_OK) return rcn; }
    const int cl = class_of_mode(e->mode[rx], bandwidth);      // DGT-U/L change chain with the bandwidth
    if (cl >= 0 && cl != e->cls[rx]) {
        const int from = e->cls[rx];
        e->cls[rx] = cl;
        int rcm = upload_members(e);
        return rcm != QK_OK ? rcm : migrate_receiver(e, rx, from, cl);
    }
    return ring_apply(e, rx);
}

int qk_set_dc_remove(qk_engine *e, int bandwidth)
{
    ENGINE_ENTER(e);
    if (bandwidth < 0) return fail(QK_EINVAL, "bad DC removal bandwidth");
    SYNC_ALL(e);
    e->dc_bw = bandwidth;
    return QK_OK;
}

int qk_set_squelch(qk_engine *e, double level_db)
{
    ENGINE_ENTER(e);
    SYNC_ALL(e);
    e->squelch_level = level_db;
    return QK_OK;
}

int qk_set_kill_audio(qk_engine *e, int on)
{
    ENGINE_ENTER(e);
    SYNC_ALL(e);
    e->kill_audio = on != 0;
    return QK_OK;
}

int qk_squelch_active(qk_engine *e, int rx)
{
    ENGINE_ENTER(e); RX_CHECK(e, rx);
    SYNC_ALL(e);
    int f = 0;
    HIPCHECK(hipMemcpy(&f, e->d_sq_flag + ((e->pipe_blocks + 1) & 1) * e->nrx + rx, sizeof(int), hipMemcpyDeviceToHost));
    return f;
}

int qk_set_noise_blanker(qk_engine *e, int level)
{
    ENGINE_ENTER(e);
    if (level < 0) return fail(QK_EINVAL, "bad noise blanker level");
    SYNC_ALL(e);
    if (level > 0 && !e->d_nb_y) {
        // hwindow_size and save_size of quisk.c:639-640; H = history carried in front of a block (>= 4 hwindow, whole segments)
        e->nb_hw = (int)(e->cfg.sample_rate * 500.E-6 + 0.5);
        e->nb_H = (4 * e->nb_hw + NB_SEG - 1) / NB_SEG * NB_SEG;
        const size_t E = (size_t)e->nb_H + e->cfg.max_block, nseg = E / NB_SEG + 2;
        for (int s = 0; s < 2; s++) {
            HIPCHECK(hipMalloc(&e->d_nb_x[s], E * sizeof(double2)));
            HIPCHECK(hipMalloc(&e->d_nb_mag[s], E * sizeof(double)));
            HIPCHECK(hipMalloc(&e->d_nb_p[s], E));
        }
        HIPCHECK(hipMalloc(&e->d_nb_y, (size_t)e->cfg.max_block * sizeof(double2)));
        HIPCHECK(hipMalloc(&e->d_nb_loc, (size_t)e->cfg.max_block * sizeof(double)));
        HIPCHECK(hipMalloc(&e->d_nb_segsum, nseg * sizeof(double)));
        HIPCHECK(hipMalloc(&e->d_nb_segbase, nseg * sizeof(double)));
        HIPCHECK(hipMalloc(&e->d_nb_S, sizeof(double)));
        HIPCHECK(hipMalloc(&e->d_nb_words, nseg * 16 * sizeof(unsigned long long)));
        HIPCHECK(hipMalloc(&e->d_nb_any, nseg));
    }
    if (level > 0 && e->nb_level <= 0) {
        // switched on: an empty delay line (the reference, whose ring simply stops moving while the blanker is off, would
        // play its stale contents first)
        const size_t E = (size_t)e->nb_H + e->cfg.max_block;
        for (int s = 0; s < 2; s++) {
            HIPCHECK(hipMemset(e->d_nb_x[s], 0, E * sizeof(double2)));
            HIPCHECK(hipMemset(e->d_nb_mag[s], 0, E * sizeof(double)));
            HIPCHECK(hipMemset(e->d_nb_p[s], 0, E));
        }
        HIPCHECK(hipMemset(e->d_nb_S, 0, sizeof(double)));
        e->nb_set = 0;
    }
    e->nb_level = level;
    return QK_OK;
}

int qk_set_invert_spectrum(qk_engine *e, int on)
{
    ENGINE_ENTER(e);
    SYNC_ALL(e);
    e->invert = on != 0;
    return QK_OK;
}

int qk_set_agc(qk_engine *e, int rx, double release_gain)
{
    ENGINE_ENTER(e); RX_CHECK(e, rx);
    SYNC_ALL(e);
    HIPCHECK(hipMemcpy(e->d_release_gain + rx, &release_gain, sizeof(double), hipMemcpyHostToDevice));
    return route_sync_params(e);
}

int qk_set_agc_release_time(qk_engine *e, double seconds)
{
    ENGINE_ENTER(e);
    if (!(seconds > 0)) return fail(QK_EINVAL, "release time must be positive");
    e->agc_release_time = seconds;
    return QK_OK;
}

int qk_process_samples_device(qk_engine *e, const double *iq_dev, int n)
{
    ENGINE_ENTER(e);
    if (!iq_dev && n > 0) return fail(QK_EINVAL, "null sample pointer");
    return process_device(e, reinterpret_cast<const double2 *>(iq_dev), n);
}

int qk_process_samples(qk_engine *e, const double *iq_host, int n)
{
    ENGINE_ENTER(e);
    if (n < 0 || n > e->cfg.max_block) return fail(QK_ERANGE, "block larger than max_block");
    if (!iq_host && n > 0) return fail(QK_EINVAL, "null sample pointer");
    if (n > 0 && e->small_mode) {
        // two pinned input blocks in turn: this one was last read by the block before last (its event: long past), so the call
        // does not wait for the previous block's tail copies and spectrum frames
        const int sl = e->in_slot ^= 1;
        double2 *dst = sl ? e->d_in_alt : e->d_in;
        if (e->in_used[sl]) HIPCHECK(hipEventSynchronize(e->ev_in[sl]));
        memcpy(dst, iq_host, (size_t)n * sizeof(double2));
        __builtin_ia32_sfence();             // (write-combined stores leave the core before the launch's doorbell)
        const int rc = process_device(e, dst, n);
        if (rc == QK_OK) { HIPCHECK(hipEventRecord(e->ev_in[sl], e->stream)); e->in_used[sl] = true; }
        else SYNC_ALL(e);
        return rc;
    }
    if (n > 0) {
        SYNC_ALL(e);      // the previous block may still read d_in
        HIPCHECK(hipMemcpyAsync(e->d_in, iq_host, (size_t)n * sizeof(double2), hipMemcpyHostToDevice, e->fe_stream));
    }
    return process_device(e, e->d_in, n);
}

int qk_process_samples_int(qk_engine *e, const void *raw, size_t n_bytes, int bytes, int big_endian)
{
    ENGINE_ENTER(e);
    if (bytes < 1 || bytes > 4) return fail(QK_EINVAL, "rx_bytes must be 1..4");
    if (n_bytes % (size_t)(2 * bytes)) return fail(QK_EINVAL, "Odd number of bytes in sample buffer");   // quisk.c:2798
    const size_t n = n_bytes / (2 * bytes);
    if (n > (size_t)e->cfg.max_block) return fail(QK_ERANGE, "buffer is too full");                       // quisk.c:2801
    if (n == 0) return process_device(e, e->d_in, 0);
    SYNC_ALL(e);
    if (n_bytes > e->raw_cap) {
        if (e->d_raw) hipFree(e->d_raw);
        e->d_raw = nullptr; e->raw_cap = 0;
        const size_t want = (size_t)e->cfg.max_block * 8;
        HIPCHECK(hipMalloc(&e->d_raw, want));
        e->raw_cap = want;                           // (only behind a successful allocation: a later call must not trust it otherwise)
    }
    HIPCHECK(hipMemcpyAsync(e->d_raw, raw, n_bytes, hipMemcpyHostToDevice, e->fe_stream));
    hipLaunchKernelGGL(k_unpack_int, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->fe_stream, e->d_raw, e->d_in, (int)n,
                       bytes, big_endian);
    return process_device(e, e->d_in, (int)n);
}

int qk_sync(qk_engine *e)
{
    ENGINE_ENTER(e);
    SYNC_ALL(e);
    AGC_ABORT_CHECK(e);
    return QK_OK;
}

int qk_audio_count(qk_engine *e, int rx)
{
    ENGINE_ENTER(e); RX_CHECK(e, rx);
    return e->out_count[rx];
}

int qk_pull_audio(qk_engine *e, int rx, double *dst, int cap)
{
    ENGINE_ENTER(e); RX_CHECK(e, rx);
    const int m = std::min(e->out_count[rx], cap);
    if (m > 0 && !dst) return fail(QK_EINVAL, "null destination");
    if (e->small_mode) {
        if (e->audio_ev) HIPCHECK(hipEventSynchronize(e->ev_audio)); else SYNC_ALL(e);
        if (m > 0) memcpy(dst, e->d_out + (size_t)rx * e->out_stride, (size_t)m * sizeof(double2));     // (the rows are host memory)
        AGC_ABORT_CHECK(e);
        return m;
    }
    HIPCHECK(hipStreamSynchronize(e->agc_stream));       // the AGC pass writes the audio output
    if (e->apply_stream) HIPCHECK(hipStreamSynchronize(e->apply_stream));
    if (m > 0) {
        HIPCHECK(hipMemcpyAsync(dst, e->d_out + (size_t)rx * e->out_stride, (size_t)m * sizeof(double2),
                                hipMemcpyDeviceToHost, e->stream));
    }
    SYNC_ALL(e);
    AGC_ABORT_CHECK(e);
    return m;
}

int qk_pull_audio_all(qk_engine *e, double *dst, size_t stride, int *counts)
{
    ENGINE_ENTER(e);
    HIPCHECK(hipStreamSynchronize(e->agc_stream));       // the AGC pass writes the audio output
    if (e->apply_stream) HIPCHECK(hipStreamSynchronize(e->apply_stream));
    if (!dst || !counts) return fail(QK_EINVAL, "null destination");
    for (int r = 0; r < e->nrx; r++) {
        const int m = e->out_count[r];
        if ((size_t)m > stride) return fail(QK_ERANGE, "stride smaller than the audio block");
        counts[r] = m;
    }
    if (e->small_mode) {
        if (e->audio_ev) HIPCHECK(hipEventSynchronize(e->ev_audio)); else SYNC_ALL(e);
        for (int r = 0; r < e->nrx; r++)
            memcpy(dst + 2 * (size_t)r * stride, e->d_out + (size_t)r * e->out_stride, std::min(stride, e->out_stride) * sizeof(double2));
        AGC_ABORT_CHECK(e);
        return QK_OK;
    }
    HIPCHECK(hipMemcpy2DAsync(dst, stride * sizeof(double2), e->d_out, e->out_stride * sizeof(double2),
                              std::min(stride, e->out_stride) * sizeof(double2), e->nrx, hipMemcpyDeviceToHost, e->stream));
    SYNC_ALL(e);
    AGC_ABORT_CHECK(e);
    return QK_OK;
}

int qk_audio_device(qk_engine *e, const double **dev_ptr, size_t *stride)
{
    ENGINE_ENTER(e);
    if (dev_ptr) *dev_ptr = reinterpret_cast<const double *>(e->d_out);
    if (stride) *stride = e->out_stride;
    return QK_OK;
}

int qk_get_graph(qk_engine *e, double zoom, double deltaf, double *out, int data_width, double *smeter)
{
    ENGINE_ENTER(e);
    if (e->fftN <= 0) return fail(QK_EINVAL, "engine created without a display FFT");
    if (!out) return fail(QK_EINVAL, "null output");
    return release_graph(e, zoom, deltaf, out, data_width, smeter);
}
int qk_graph_pending(qk_engine *e) { ENGINE_ENTER(e); return e->fft_count; }
int qk_graph_mark(qk_engine *e, double zoom, double deltaf)
{
    ENGINE_ENTER(e);
    if (e->fftN <= 0) return fail(QK_EINVAL, "engine created without a display FFT");
    if (e->rel_pending) return fail(QK_EINVAL, "a marked row has not been collected yet (qk_get_graph)");
    return graph_enqueue(e, zoom, deltaf);
}
int qk_graph_device(qk_engine *e, const double **dev_ptr) { ENGINE_ENTER(e); if (dev_ptr) *dev_ptr = e->d_graph; return QK_OK; }
double qk_get_smeter(qk_engine *e) { return e ? e->smeter : 0.0; }

int qk_multirx_graph(qk_engine *e, const double *iq_host, int data_width, double *out)
{
    ENGINE_ENTER(e);
    if (!iq_host || !out || data_width < 1 || data_width > (1 << 20)) return fail(QK_EINVAL, "bad multirx graph arguments");
    const int mult = 8;                                  // MULTIRX_FFT_MULT, quisk.c:32
    const int N = data_width * mult;                     // multirx_fft_width, quisk.c:5098
    if (e->mrx_width != data_width) {
        HIPCHECK(hipStreamSynchronize(e->fe_stream));
        any_free(e->mrx);
        for (void *p : {(void *)e->d_mrx_in, (void *)e->d_mrx_X, (void *)e->d_mrx_out}) if (p) hipFree(p);
        e->d_mrx_in = e->d_mrx_X = nullptr; e->d_mrx_out = nullptr; e->mrx_width = 0;
        std::vector<double> win(N);
        for (int i = 0, j = -N / 2; i < N; i++, j++) win[i] = 0.5 + 0.5 * cos(2. * M_PI * j / N);     // quisk.c:4481-4482
        int rc = any_init(e, e->mrx, N, win.data());
        if (rc != QK_OK) return rc;
        HIPCHECK(hipMalloc(&e->d_mrx_in, sizeof(double2) * N));
        HIPCHECK(hipMalloc(&e->d_mrx_X, sizeof(double2) * N));
        HIPCHECK(hipMalloc(&e->d_mrx_out, sizeof(double) * data_width));
        e->mrx_width = data_width;
    }
    hipStream_t st = e->fe_stream;
    HIPCHECK(hipMemcpyAsync(e->d_mrx_in, iq_host, sizeof(double2) * N, hipMemcpyHostToDevice, st));
    any_run(e, e->mrx, st, e->d_mrx_in, nullptr, e->d_mrx_X);
    const double scale = 20.0 * (log10((double)N) + 31.0 * log10(2.0));      // quisk.c:4493-4494
    hipLaunchKernelGGL(k_multirx_bins, dim3((data_width + 255) / 256), dim3(256), 0, st, e->d_mrx_X, N, mult, data_width, scale, e->d_mrx_out);
    HIPCHECK(hipMemcpyAsync(out, e->d_mrx_out, sizeof(double) * data_width, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    HIPCHECK(hipGetLastError());
    return QK_OK;
}

int qk_bandscope_init(qk_engine *e, int size)
{
    ENGINE_ENTER(e);
    if (size < 2 || size > (1 << 22)) return fail(QK_EINVAL, "bad bandscope size");
    HIPCHECK(hipStreamSynchronize(e->fe_stream));
    any_free(e->bs);
    for (void *p : {(void *)e->d_bs_in, (void *)e->d_bs_avg, (void *)e->d_bs_X, (void *)e->d_bs_max}) if (p) hipFree(p);
    e->d_bs_in = e->d_bs_avg = nullptr; e->d_bs_X = nullptr; e->d_bs_max = nullptr; e->bs_size = 0; e->bs_count = 0;
    std::vector<double> win(size);
    for (int i = 0, j = -size / 2; i < size; i++, j++) win[i] = 0.5 + 0.5 * cos(2. * M_PI * j / size);   // quisk.c:2769-2770
    int rc = any_init(e, e->bs, size, win.data());
    if (rc != QK_OK) return rc;
    const int L = size / 2 + 1;
    HIPCHECK(hipMalloc(&e->d_bs_in, sizeof(double) * size));
    HIPCHECK(hipMalloc(&e->d_bs_X, sizeof(double2) * size));
    HIPCHECK(hipMalloc(&e->d_bs_avg, sizeof(double) * (L + 1)));
    HIPCHECK(hipMemset(e->d_bs_avg, 0, sizeof(double) * (L + 1)));
    HIPCHECK(hipMalloc(&e->d_bs_max, sizeof(unsigned long long)));
    HIPCHECK(hipMemset(e->d_bs_max, 0, sizeof(unsigned long long)));
    e->bs_size = size;
    return QK_OK;
}

int qk_add_bscope_samples(qk_engine *e, const double *samples_host, int n)
{
    ENGINE_ENTER(e);
    if (e->bs_size <= 0) return fail(QK_EINVAL, "The bandscope was not initialized");        // quisk.c:2851-2855
    if (!samples_host || n != e->bs_size) return fail(QK_EINVAL, "Wrong number of samples in bandscope block");
    hipStream_t st = e->fe_stream;
    const int L = n / 2 + 1;
    HIPCHECK(hipMemcpyAsync(e->d_bs_in, samples_host, sizeof(double) * n, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_absmax, dim3(std::min(64, (n + 255) / 256)), dim3(256), 0, st, e->d_bs_in, n, e->d_bs_max);
    any_run(e, e->bs, st, nullptr, e->d_bs_in, e->d_bs_X);
    hipLaunchKernelGGL(k_bscope_accum, dim3((L + 255) / 256), dim3(256), 0, st, e->d_bs_X, L, e->d_bs_avg);
    HIPCHECK(hipStreamSynchronize(st));          // the host buffer may be reused at once, as in the reference
    HIPCHECK(hipGetLastError());
    e->bs_count++;
    return QK_OK;
}

int qk_add_bscope_samples_int(qk_engine *e, const void *raw, size_t n_bytes, int bytes, int big_endian)
{
    if (!e) return fail(QK_EINVAL, "null engine");
    if (bytes < 1 || bytes > 4 || !raw) return fail(QK_EINVAL, "bad bandscope sample format");
    if (e->bs_size <= 0) return fail(QK_EINVAL, "The bandscope was not initialized");
    if (n_bytes != (size_t)e->bs_size * bytes) return fail(QK_EINVAL, "Wrong number of bytes in sample buffer");   // quisk.c:2867
    // a few KiB per block: unpacked on the host (left-justified into int32, then / CLIP32, quisk.c:2870-2895)
    std::vector<double> v((size_t)e->bs_size);
    const unsigned char *p = static_cast<const unsigned char *>(raw);
    for (int i = 0; i < e->bs_size; i++, p += bytes) {
        uint32_t u = 0;
        for (int b = 0; b < bytes; b++) {
            const int sig = big_endian ? (bytes - 1 - b) : b;
            u |= (uint32_t)p[b] << (8 * (4 - bytes + sig));
        }
        v[i] = (double)(int32_t)u / QK_CLIP32;
    }
    return qk_add_bscope_samples(e, v.data(), e->bs_size);
}

int qk_get_bandscope(qk_engine *e, int clock, double zoom, double deltaf, double *out, int graph_width, double *adc_level)
{
    ENGINE_ENTER(e);
    if (e->bs_size <= 0) return fail(QK_EINVAL, "The bandscope was not initialized");
    if (!out || graph_width < 1 || clock <= 0) return fail(QK_EINVAL, "bad bandscope arguments");
    const int count = e->bs_count;
    if (count <= 0) return 0;
    hipStream_t st = e->fe_stream;
    if (graph_width > e->bs_out_cap) {
        HIPCHECK(hipStreamSynchronize(st));
        if (e->d_bs_out) hipFree(e->d_bs_out);
        e->d_bs_out = nullptr; e->bs_out_cap = 0;
        HIPCHECK(hipMalloc(&e->d_bs_out, sizeof(double) * graph_width));
        e->bs_out_cap = graph_width;
    }
    const int L = e->bs_size / 2 + 1;
    const double frac = (double)L / graph_width;
    BscopeRelease a{e->d_bs_avg, e->d_bs_out, L, graph_width, zoom, deltaf, clock / 2.0, 1.0 / frac / count / e->bs_size};
    hipLaunchKernelGGL(k_bscope_release, dim3((graph_width + 255) / 256), dim3(256), 0, st, a);
    unsigned long long mx = 0;
    HIPCHECK(hipMemcpyAsync(out, e->d_bs_out, sizeof(double) * graph_width, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipMemcpyAsync(&mx, e->d_bs_max, sizeof mx, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipMemsetAsync(e->d_bs_avg, 0, sizeof(double) * (L + 1), st));
    HIPCHECK(hipMemsetAsync(e->d_bs_max, 0, sizeof mx, st));
    HIPCHECK(hipStreamSynchronize(st));
    if (adc_level) { double d; memcpy(&d, &mx, sizeof d); *adc_level = d; }     // hermes_adc_level, quisk.c:4591
    e->bs_count = 0;
    return count;
}

int qk_set_ssb_squelch(qk_engine *e, int enabled, int level)
{
    ENGINE_ENTER(e);
    SYNC_ALL(e);
    if (e->nrx > 0 && !enabled && e->ssb_sq_enabled) {      // receivers the SSB squelch had closed open again
        std::vector<int> fl(2 * (size_t)e->nrx);
        HIPCHECK(hipMemcpy(fl.data(), e->d_sq_flag, fl.size() * sizeof(int), hipMemcpyDeviceToHost));
        for (int b = 0; b < 2; b++) for (int r = 0; r < e->nrx; r++) if (e->cls[r] != CL_FM) fl[(size_t)b * e->nrx + r] = 0;
        HIPCHECK(hipMemcpy(e->d_sq_flag, fl.data(), fl.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    e->ssb_sq_enabled = enabled != 0; e->ssb_sq_level = level;
    return QK_OK;
}

static int notch_reset(qk_engine *e, int rx)
{   // dAutoNotch(NULL, ...), quisk.c:761-770
    if (!e->d_notch) return QK_OK;
    HIPCHECK(hipMemset(e->d_notch + rx, 0, sizeof(NotchState)));
    const int init[8] = {AN_START, 0, -4, 0, -4, -1, 0, 0};     // index, old1, count1, old2, count2, fltrSig
    HIPCHECK(hipMemcpy(e->d_notch + rx, init, sizeof init, hipMemcpyHostToDevice));
    return QK_OK;
}

int qk_set_auto_notch(qk_engine *e, int rx, int on)
{
    ENGINE_ENTER(e); RX_CHECK(e, rx);
    SYNC_ALL(e);
    if (on && !e->d_notch) HIPCHECK(hipMalloc(&e->d_notch, sizeof(NotchState) * (size_t)e->nrx));
    on = on != 0;
    e->notch_on[rx] = on;
    HIPCHECK(hipMemcpy(e->d_notch_on + rx, &on, sizeof(int), hipMemcpyHostToDevice));
    e->any_notch = false;
    for (int v : e->notch_on) e->any_notch = e->any_notch || v;
    return notch_reset(e, rx);
}

int qk_set_rit_freq(qk_engine *e, int hz)
{
    ENGINE_ENTER(e);
    SYNC_ALL(e);
    e->rit_freq = hz;
    for (int r = 0; r < e->nrx; r++)                        // for CW, changing the RIT affects the notch (quisk.c:4343)
        if (e->notch_on[r]) { int rc = notch_reset(e, r); if (rc != QK_OK) return rc; }
    return QK_OK;
}

int qk_is_squelch(qk_engine *e, int freq)
{   // IsSquelch, quisk.c:1231-1252, on the graph row of the last release (current_graph, quisk.c:4747)
    ENGINE_ENTER(e);
    if (e->fftN <= 0 || !e->h_graph) return fail(QK_EINVAL, "engine created without a display FFT");
    const int W = e->fftW, fs = e->cfg.sample_rate;
    int bw = 5000 * W / fs;
    if (bw < 1) bw = 1;
    const int i1 = (int)((double)freq * W / fs + W / 2.0 - bw / 2.0 + 0.5), i2 = i1 + bw;
    double meter = 0;
    if (i1 >= 0 && i2 < W && e->graph_rows > 0)
        for (int i = i1; i < i2; i++) meter += e->last_row[i];
    meter /= bw;
    return (meter == 0 || meter < e->squelch_level) ? 1 : 0;
}

// ---- stereo routes ----
static int build_route_tail(qk_engine *e, qk_engine::Route &r)
{
    Chain &t = r.tail;
    const Stream &dec = e->front.streams[e->front.aud];
    const int cap_in = dec.cap + e->route_cap + 64;
    ChainBuilder b(t, cap_in);
    b.cur = b.new_stream(1, cap_in);                                   // stream 0: the pair (row 0 = left, row 1 = right)
    b.cur_cap = cap_in;
    if (e->decim_srate != 48000) {                                     // quisk.c:2528-2532
        b.need_hist(b.cur, 3);
        Stage &st = b.add(ST_FRAC, 1, b.cur_cap + 2);
        st.timer = QK_KT_AUDIO; st.obj = OB_FRAC;
        t.frac_cap = b.cur_cap + 8;
        HIPCHECK(hipMalloc(&t.d_frac_idx, sizeof(int) * t.frac_cap));
        HIPCHECK(hipMalloc(&t.d_frac_pos, sizeof(double) * t.frac_cap));
        for (auto &fs : t.frac_slot) {
            HIPCHECK(hipHostMalloc(&fs.idx, sizeof(int) * t.frac_cap, hipHostMallocMapped));
            HIPCHECK(hipHostMalloc(&fs.pos, sizeof(double) * t.frac_cap, hipHostMallocMapped));
            HIPCHECK(hipEventCreateWithFlags(&fs.ev, sync_event_flags()));
        }
        t.frac_dindex = 1.0;
    }
    for (int q = e->cfg.playback_rate / 48000, k = 0; q > 1; q /= 2, k++) {     // quisk.c:2535-2553
        b.hbint(1, QK_KT_AUDIO);
        t.stages.back().obj = OB_PHB0 + k;
    }
    t.play = t.aud = b.cur;
    b.need_hist(b.cur, e->cfg.playback_rate * 15 / 1000);             // AGC_DELAY, quisk.c:30, 2036
    int rc = alloc_chain_streams(e, t, 2);
    if (rc != QK_OK) return rc;
    Stream &pl = t.streams[t.play];
    const size_t bytes = (size_t)2 * pl.rx_stride() * sizeof(double);
    HIPCHECK(hipMalloc(&t.play_alt, bytes));
    HIPCHECK(hipMemsetAsync(t.play_alt, 0, bytes, e->stream));
    HIPCHECK(hipStreamSynchronize(e->stream));
    t.built = true;
    return QK_OK;
}

// the route's AGC slots follow the receiver's settings: agcReleaseGain and rxMode are globals in the reference (quisk.c:145, 2088)
static int route_sync_params(qk_engine *e)
{
    for (auto &r : e->routes) {
        if (!r.tail.built) continue;
        for (int ch = 0; ch < 2; ch++) {
            HIPCHECK(hipMemcpy(e->d_mode + r.slot + ch, e->d_mode + r.rx, sizeof(int), hipMemcpyDeviceToDevice));
            HIPCHECK(hipMemcpy(e->d_release_gain + r.slot + ch, e->d_release_gain + r.rx, sizeof(double), hipMemcpyDeviceToDevice));
        }
    }
    return QK_OK;
}

// A route goes live or stops (all streams idle).  What the reference keeps in ONE place for mono and stereo alike moves with
// it: cFracDecim's statics and HalfBand7..9 hold the last samples of `d + I d` when the split starts -- both channels continue
// from the mono history -- and Agc1 (state and delay ring) is the left channel's AGC (quisk.c:2178-2181, 2528-2575).  Agc2
// (slot + 1) keeps its state from one stretch of stereo to the next.  Going back to mono the left channel's objects return.
static int route_transfer(qk_engine *e, qk_engine::Route &r, bool to_live)
{
    Chain &c = e->cls_chain[e->cls[r.rx]], &t = r.tail;
    for (auto &ts : t.stages) {
        if (ts.obj < 0) continue;
        for (auto &cs : c.stages) {
            if (cs.obj != ts.obj) continue;
            const int need = obj_need_of(e, ts.obj);
            double *cp = hist_tail(e, c, cs.in, r.rx, 0, need);
            for (int ch = 0; ch < (to_live ? 2 : 1); ch++) {
                double *tp = hist_tail(e, t, ts.in, ch, 0, need);
                HIPCHECK(hipMemcpy(to_live ? tp : cp, to_live ? cp : tp, sizeof(double) * need, hipMemcpyDeviceToDevice));
            }
            if (ts.kind == ST_FRAC) {
                if (to_live) t.frac_dindex = c.frac_dindex;
                else if (c.members.size() == 1) c.frac_dindex = t.frac_dindex;
            }
        }
    }
    // Agc1: state, "has run", delay ring (the play stream's history, in both twins)
    const int from = to_live ? r.rx : r.slot, to = to_live ? r.slot : r.rx;
    HIPCHECK(hipMemcpy(e->d_agc + to, e->d_agc + from, sizeof(AgcState), hipMemcpyDeviceToDevice));
    e->agc_started[to] = e->agc_started[from];
    const int flag = e->agc_started[to];
    HIPCHECK(hipMemcpy(e->d_agc_started + to, &flag, sizeof(int), hipMemcpyHostToDevice));
    if (!to_live && e->d_agc_started2) HIPCHECK(hipMemcpy(e->d_agc_started2 + to, &flag, sizeof(int), hipMemcpyHostToDevice));
    if (c.built && e->agc_started[to]) {
        Stream &cp = c.streams[c.play], &tp = t.streams[t.play];
        const int h = std::min(cp.hist, tp.hist);
        double *cbase = c.play_alt + (long long)r.rx * cp.rx_stride() + (cp.hist - h);       // (plane 0: mono; the left channel of DGT-IQ)
        double *cbase2 = cp.d + (long long)r.rx * cp.rx_stride() + (cp.hist - h);
        double *tbase = t.play_alt + (tp.hist - h), *tbase2 = tp.d + (tp.hist - h);
        if (to_live) {
            HIPCHECK(hipMemcpy(tbase, cbase, sizeof(double) * h, hipMemcpyDeviceToDevice));
            HIPCHECK(hipMemcpy(tbase2, cbase, sizeof(double) * h, hipMemcpyDeviceToDevice));
        } else {
            HIPCHECK(hipMemcpy(cbase, tbase, sizeof(double) * h, hipMemcpyDeviceToDevice));
            HIPCHECK(hipMemcpy(cbase2, tbase, sizeof(double) * h, hipMemcpyDeviceToDevice));
        }
    }
    return QK_OK;
}

// The right channel's AGC ring.  Stereo stops: Agc2 keeps its ring for the next stretch (d_agc2_ring), and the play stream's
// right row -- from now on the imaginary part of Agc1's ring -- starts as zeros: the stereo path stored real samples there
// (quisk.c:2564-2568).  Stereo starts: Agc2's ring comes back.  The latest history is in the twin that is written next.
static int route_stereo(qk_engine *e, qk_engine::Route &r, bool on)
{
    Chain &t = r.tail;
    Stream &p = t.streams[t.play];
    const int bs = e->cfg.playback_rate * 15 / 1000;
    if (!r.d_agc2_ring) {
        HIPCHECK(hipMalloc(&r.d_agc2_ring, sizeof(double) * bs));
        HIPCHECK(hipMemset(r.d_agc2_ring, 0, sizeof(double) * bs));
    }
    double *row1[2] = {t.play_alt + p.rx_stride() + (p.hist - bs), p.d + p.rx_stride() + (p.hist - bs)};
    if (on) {
        for (double *d : row1) HIPCHECK(hipMemcpy(d, r.d_agc2_ring, sizeof(double) * bs, hipMemcpyDeviceToDevice));
    } else {
        HIPCHECK(hipMemcpy(r.d_agc2_ring, row1[0], sizeof(double) * bs, hipMemcpyDeviceToDevice));
        for (double *d : row1) HIPCHECK(hipMemset(d, 0, sizeof(double) * bs));
        // how long the two channels can still differ, in output samples: cFracDecim (3 inputs) and the half-band interpolators
        // (22 inputs each, at doubling rates) -- under 42 samples at the decimated rate -- and then Agc1's ring
        r.drain = bs + 42LL * (e->cfg.playback_rate / 48000) + 8;
    }
    r.stereo = on;
    return QK_OK;
}

// bring every route in line with what was asked for: it is stereo while its method is 1..4 and its receiver is not in DGT-IQ
// (which is stereo by itself: the reference leaves the second channel alone then, quisk.c:2407-2409); behind a stretch of
// stereo it runs on in mono until the channels agree again (process_device ends that)
static int routes_refresh(qk_engine *e)
{
    bool changed = false;
    for (auto &r : e->routes) {
        const bool want = r.method > 0 && e->cls[r.rx] != CL_DGTIQ;
        if (want == r.stereo) continue;
        if (want) {
            if (!r.tail.built) { int rc = build_route_tail(e, r); if (rc != QK_OK) return rc; }
            if (!r.live) {
                int rc = route_transfer(e, r, true);
                if (rc != QK_OK) return rc;
                r.live = true; changed = true;
            }
        }
        int rc = route_stereo(e, r, want);
        if (rc != QK_OK) return rc;
    }
    if (changed) { int rc = upload_agc_cls(e); if (rc != QK_OK) return rc; }
    return route_sync_params(e);
}

int qk_set_stereo_route(qk_engine *e, int rx, int partner_rx, int method)
{
    ENGINE_ENTER(e); RX_CHECK(e, rx);
    if (method < 0 || method > 4) return fail(QK_EINVAL, "route method must be 0..4");
    if (method > 0) { RX_CHECK(e, partner_rx); if (partner_rx == rx) return fail(QK_EINVAL, "a receiver cannot be its own partner"); }
    SYNC_ALL(e);
    qk_engine::Route *rt = nullptr;
    for (auto &r : e->routes) if (r.rx == rx) rt = &r;
    if (!rt) {
        if (method == 0) return QK_OK;
        if ((int)e->routes.size() >= QK_MAX_ROUTES) return fail(QK_ERANGE, "too many stereo routes (QK_MAX_ROUTES)");
        // allocate first, commit to the engine only when everything is there
        qk_engine::Route nr;
        nr.rx = rx; nr.slot = e->nrx + 2 * (int)e->routes.size();
        const size_t fifo_len = (size_t)e->route_cap + e->front.streams[e->front.aud].cap + 64;
        bool ok = true;
        for (auto &q : nr.fifo) for (double *&p : q) ok = ok && hipMalloc(&p, sizeof(double) * fifo_len) == hipSuccess;
        if (!ok) {
            for (auto &q : nr.fifo) for (double *p : q) if (p) hipFree(p);
            return fail(QK_ENOMEM, "out of device memory for a route's FIFOs");
        }
        e->routes.push_back(nr);
        rt = &e->routes.back();
    }
    if (rt->method == 0 || method == 0 || rt->partner != partner_rx) rt->nbuf_a = rt->nbuf_b = 0;      // Buffer2Chan(NULL, ...): initialise (quisk.c:2227-2231)
    if (method > 0) rt->partner = partner_rx;
    rt->method = method;
    return routes_refresh(e);
}

int qk_key_up(qk_engine *e)
{
    ENGINE_ENTER(e);
    e->keyup_env = 0.0;              // quisk.c:2241
    return QK_OK;
}

int qk_change_scan(qk_engine *e, int scan_blocks, int scan_sample_rate, double scan_valid, int scan_vfo0, int scan_deltaf,
                   int tx_tune_hz, int vfo_screen)
{
    ENGINE_ENTER(e);
    if (e->fftN <= 0) return fail(QK_EINVAL, "engine created without a display FFT");
    if (scan_blocks < 0 || (scan_blocks > 0 && (scan_sample_rate <= 0 || !(scan_valid > 0) || scan_valid > 1)))
        return fail(QK_EINVAL, "bad scan parameters");
    HIPCHECK(hipStreamSynchronize(e->fe_stream));
    if (scan_blocks > 0 && !e->d_scan_avg) HIPCHECK(hipMalloc(&e->d_scan_avg, sizeof(double) * e->fftN));
    if (scan_blocks > 0) HIPCHECK(hipMemset(e->d_scan_avg, 0, sizeof(double) * e->fftN));
    if (e->fftN1 <= 1) HIPCHECK(hipMemset(e->d_fft_avg, 0, sizeof(double) * e->fftN));
    e->scan_blocks = scan_blocks; e->scan_rate = scan_sample_rate; e->scan_valid = scan_valid; e->scan_vfo0 = scan_vfo0;
    e->scan_deltaf = scan_deltaf; e->scan_tx_tune = tx_tune_hz; e->scan_vfo_screen = vfo_screen; e->scan_next = 0;
    e->fft_count = 0; e->fft_parts = 0;
    if (e->rel_pending) { hipEventSynchronize(e->ev_graph); e->rel_pending = false; }    // a marked row of the old setting is dropped
    return QK_OK;
}

int qk_set_graph_raw(qk_engine *e, int on)
{
    ENGINE_ENTER(e);
    if (e->fftN <= 0) return fail(QK_EINVAL, "engine created without a display FFT");
    on = on != 0;
    if (on == e->raw_mode) return QK_OK;
    HIPCHECK(hipStreamSynchronize(e->fe_stream));
    if (on && !e->d_raw_rows) HIPCHECK(hipMalloc(&e->d_raw_rows, sizeof(double2) * 4 * e->fftW));
    e->raw_mode = on; e->raw_head = e->raw_count = 0; e->fft_fill = 0;
    e->fft_count = 0; e->fft_parts = 0;
    if (e->rel_pending) { hipEventSynchronize(e->ev_graph); e->rel_pending = false; }    // a marked row of the old setting is dropped                                  // count_fft = 0, quisk.c:4618-4621
    if (e->fftN1 <= 1) HIPCHECK(hipMemset(e->d_fft_avg, 0, sizeof(double) * e->fftN));
    HIPCHECK(hipMemset(e->d_meter, 0, sizeof(double)));
    return QK_OK;
}

int qk_get_graph_raw(qk_engine *e, double *out, int data_width)
{
    ENGINE_ENTER(e);
    if (!e->raw_mode) return fail(QK_EINVAL, "raw scope is off (qk_set_graph_raw)");
    if (!out || data_width != e->fftW) return fail(QK_EINVAL, "data_width differs from the engine's");
    if (e->raw_count <= 0) return 0;
    HIPCHECK(hipMemcpyAsync(out, e->d_raw_rows + (size_t)e->raw_head * e->fftW, sizeof(double2) * e->fftW, hipMemcpyDeviceToHost, e->fe_stream));
    HIPCHECK(hipStreamSynchronize(e->fe_stream));
    e->raw_head = (e->raw_head + 1) & 3; e->raw_count--;
    return 1;
}

int qk_rx_filter_block(qk_engine *e, int rx, int bank, const double *in_host, double *out_host, int n)
{
    ENGINE_ENTER(e); RX_CHECK(e, rx);
    if (bank < 0 || bank > 2 || n < 0 || n > e->cfg.max_block || (n > 0 && (!in_host || !out_host))) return fail(QK_EINVAL, "bad filter block arguments");
    if (n == 0) return 0;
    const int H = e->tap_cap, nt = e->ntaps[rx];
    if (!e->d_rfb_y) {
        HIPCHECK(hipMalloc(&e->d_rfb_y, sizeof(double2) * e->cfg.max_block));
        HIPCHECK(hipMalloc(&e->d_rfb_t, sizeof(double2) * H));
    }
    if (!e->d_rfb_x[bank]) {
        HIPCHECK(hipMalloc(&e->d_rfb_x[bank], sizeof(double2) * ((size_t)H + e->cfg.max_block)));
        HIPCHECK(hipMemset(e->d_rfb_x[bank], 0, sizeof(double2) * H));
    }
    hipStream_t st = e->stream;
    double2 *x = e->d_rfb_x[bank];
    HIPCHECK(hipMemcpyAsync(x + H, in_host, sizeof(double2) * n, hipMemcpyHostToDevice, st));
    if (nt > 0) {
        const double *hI = e->d_taps + (size_t)rx * 2 * e->tap_cap;
        hipLaunchKernelGGL(k_rxf_block, dim3((n + 255) / 256), dim3(256), 0, st, x, e->d_rfb_y, hI, hI + e->tap_cap, nt, H, n);
        HIPCHECK(hipMemcpyAsync(out_host, e->d_rfb_y, sizeof(double2) * n, hipMemcpyDeviceToHost, st));
    }
    // the last H samples become the history of the next call
    hipLaunchKernelGGL(k_copy_c, dim3((H + 255) / 256), dim3(256), 0, st, x + n, e->d_rfb_t, H);
    hipLaunchKernelGGL(k_copy_c, dim3((H + 255) / 256), dim3(256), 0, st, e->d_rfb_t, x, H);
    HIPCHECK(hipStreamSynchronize(st));
    HIPCHECK(hipGetLastError());
    if (nt <= 0) memcpy(out_host, in_host, sizeof(double2) * n);       // sizeFilter == 0: the sample passes (quisk.c:1164-1165)
    return n;
}

int qk_add_tone(qk_engine *e, int freq_hz)
{   // add_tone, quisk.c:3022-3034: testtonePhase = cexp(I 2 pi freq / rate), 0 = off.  The vectors are statics of AddTestTone
    // (testtoneVector = 21474836.47, audioVector = 1, quisk.c:1197-1198): they carry on from one tone to the next
    ENGINE_ENTER(e);
    SYNC_ALL(e);
    const int nA = (e->cfg.max_block >> 12) + 2;
    if (e->nrx == 0) e->nA = nA;
    if (freq_hz != 0 && !e->d_tone_tab) {
        HIPCHECK(hipMalloc(&e->d_tone_tab, sizeof(double2) * (size_t)(2 * nA + 256)));
        HIPCHECK(hipMalloc(&e->d_tone_buf, sizeof(double2) * (size_t)e->cfg.max_block));
        const double2 init[2] = {make_double2(21474836.47, 0.0), make_double2(1.0, 0.0)};
        for (int i = 0; i < 2; i++) {
            HIPCHECK(hipMalloc(&e->d_tone_state[i], sizeof(double2) * 2));
            HIPCHECK(hipMemcpy(e->d_tone_state[i], init, sizeof init, hipMemcpyHostToDevice));
        }
        // the audio vector's step: cexp(I 2 pi 1000 / rate) (quisk.c:1211, 1221)
        double pre, pim;
        qk_phase_step(-1000, e->cfg.sample_rate, &pre, &pim);
        std::vector<double> C(128), B(128), A((size_t)2 * nA);
        qk_phase_tables(pre, pim, C.data(), B.data(), A.data(), nA);
        HIPCHECK(hipMemcpy(e->d_tone_tab + nA + 128, A.data(), sizeof(double) * 2 * nA, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(e->d_tone_tab + 2 * nA + 128, B.data(), sizeof(double) * 128, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(e->d_tone_tab + 2 * nA + 192, C.data(), sizeof(double) * 128, hipMemcpyHostToDevice));
    }
    if (freq_hz != 0) {
        double pre, pim;
        qk_phase_step(-freq_hz, e->cfg.sample_rate, &pre, &pim);       // = cexp((I * 2.0 * M_PI * freq) / rate), quisk.c:3030
        std::vector<double> C(128), B(128), A((size_t)2 * nA);
        qk_phase_tables(pre, pim, C.data(), B.data(), A.data(), nA);
        HIPCHECK(hipMemcpy(e->d_tone_tab, A.data(), sizeof(double) * 2 * nA, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(e->d_tone_tab + nA, B.data(), sizeof(double) * 128, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(e->d_tone_tab + nA + 64, C.data(), sizeof(double) * 128, hipMemcpyHostToDevice));
    }
    e->tone_freq = freq_hz;
    return QK_OK;
}

// one transform of n points through the engine's FFT: forward, or FFTW's unnormalised backward transform as
// conj(forward(conj(x))).  in / out: n interleaved complex doubles in host memory.
static int fft_any_host(qk_engine *e, const double *in, int n, int inverse, double *out)
{
    if (n == 1) { out[0] = in[0]; out[1] = in[1]; return QK_OK; }
    if (e->gen_n != n) {
        HIPCHECK(hipStreamSynchronize(e->fe_stream));
        any_free(e->gen);
        for (void *p : {(void *)e->d_gen_in, (void *)e->d_gen_out}) if (p) hipFree(p);
        e->d_gen_in = e->d_gen_out = nullptr; e->gen_n = 0;
        int rc = any_init(e, e->gen, n, nullptr);
        if (rc != QK_OK) return rc;
        HIPCHECK(hipMalloc(&e->d_gen_in, sizeof(double2) * n));
        HIPCHECK(hipMalloc(&e->d_gen_out, sizeof(double2) * n));
        e->gen_n = n;
    }
    std::vector<double> tmp;
    const double *src = in;
    if (inverse) {
        tmp.assign(in, in + 2 * (size_t)n);
        for (int i = 0; i < n; i++) tmp[2 * i + 1] = -tmp[2 * i + 1];
        src = tmp.data();
    }
    hipStream_t st = e->fe_stream;
    HIPCHECK(hipMemcpyAsync(e->d_gen_in, src, sizeof(double2) * n, hipMemcpyHostToDevice, st));
    any_run(e, e->gen, st, e->d_gen_in, nullptr, e->d_gen_out);
    HIPCHECK(hipMemcpyAsync(out, e->d_gen_out, sizeof(double2) * n, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    HIPCHECK(hipGetLastError());
    if (inverse) for (int i = 0; i < n; i++) out[2 * i + 1] = -out[2 * i + 1];
    return QK_OK;
}

int qk_measure_audio(qk_engine *e, int time, double *value)
{   // measure_audio, quisk.c:3011-3020: time > 0 sets the averaging time in seconds; returns measured_audio (RMS of receiver 0's
    // detected audio over that time, in millionths of full scale)
    ENGINE_ENTER(e);
    if (time > 0) e->ma_time = time;
    double v[2] = {0, 0};
    if (e->d_ma) {
        HIPCHECK(hipStreamSynchronize(e->stream));
        if (e->apply_stream) HIPCHECK(hipStreamSynchronize(e->apply_stream));
        HIPCHECK(hipStreamSynchronize(e->agc_stream));
        HIPCHECK(hipMemcpy(v, e->d_ma, sizeof v, hipMemcpyDeviceToHost));
    }
    if (value) *value = v[1];
    return QK_OK;
}

int qk_measure_frequency(qk_engine *e, int mode, double *value)
{   // measure_frequency, quisk.c:3000-3009: mode >= 0 sets measure_freq_mode (0 = off; a measurement every mode / 2 frames of two
    // seconds), mode < 0 only reads; returns measured_frequency in Hz from the centre of the stream
    ENGINE_ENTER(e);
    if (mode >= 0 && mode != e->mf_mode) { SYNC_ALL(e); e->mf_mode = mode; }
    if (value) *value = e->mf_value;
    return QK_OK;
}

int qk_get_audio_graph(qk_engine *e, double *out, int data_width)
{   // get_audio_graph, quisk.c:966-991: 1 and data_width dB values when a graph is ready (it is then cleared), else 0
    ENGINE_ENTER(e);
    if (!out || data_width != e->cfg.data_width) return fail(QK_EINVAL, "data_width differs from the engine's");
    if (!e->ag_ready) return 0;
    const int N = e->ag_N;
    SYNC_ALL(e);
    HIPCHECK(hipMemcpy(out, e->d_ag_avg, sizeof(double) * N, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemset(e->d_ag_avg, 0, sizeof(double) * N));
    for (int i = 0; i < N; i++) out[i] = 20.0 * log10(out[i] < 1E-10 ? 1E-10 : out[i]);
    e->ag_ready = 0;
    return 1;
}

int qk_fft_any(qk_engine *e, const double *in_host, int n, int inverse, double *out_host)
{
    ENGINE_ENTER(e);
    if (!in_host || !out_host || n < 1 || n > (1 << 22)) return fail(QK_EINVAL, "bad transform arguments");
    return fft_any_host(e, in_host, n, inverse, out_host);
}

int qk_get_filter(qk_engine *e, double *out, int data_width)
{   // get_filter, quisk.c:4761-4848: the response of receiver 0's band-pass (filter slot 0) to a comb of cosines at every bin
    // frequency up to data_width / 2 - 10, through the reference's ring in its tap order, the first data_width entries of the
    // DISPLAY window (fft_window has fft_size entries, quisk.c:4816-4817), a data_width-point transform, 20 log10(|X| / width)
    ENGINE_ENTER(e);
    const int W = data_width, N = (int)e->f0_i.size();
    if (!out || W < 2 || W > (1 << 20)) return fail(QK_EINVAL, "bad get_filter arguments");
    if (e->cfg.fft_size < W) return fail(QK_EINVAL, "get_filter needs the display window (engine made with fft_size >= data_width)");
    const int T = W + N;
    std::vector<double> avg((size_t)T, 0.5);                          // Value for freq == 0
    for (int freq = 1; freq < W / 2.0 - 10.0; freq++) {
        const double delta = 2 * M_PI / W * freq;
        double phase = 0;
        for (int t = 0; t < T; t++) {
            avg[t] += cos(phase);
            phase += delta;
            if (phase > 2 * M_PI) phase -= 2 * M_PI;
        }
    }
    std::vector<double> bufI((size_t)std::max(N, 1), 0.0), bufQ((size_t)std::max(N, 1), 0.0), sam((size_t)2 * W, 0.0), X((size_t)2 * W);
    int nidx = 0;
    for (int t = 0; t < T; t++) {
        double accI = 0, accQ = 0;
        if (N > 0) {
            bufI[nidx] = avg[t]; bufQ[nidx] = avg[t];
            int j = nidx;
            for (int k = 0; k < N; k++) {
                accI += bufI[j] * e->f0_i[k];
                accQ += bufQ[j] * e->f0_q[k];
                if (++j >= N) j = 0;
            }
            if (++nidx >= N) nidx = 0;
        }
        if (t >= N) { sam[2 * (size_t)(t - N)] = accI; sam[2 * (size_t)(t - N) + 1] = accQ; }
    }
    const int F = e->cfg.fft_size;
    for (int i = 0; i < W; i++) {
        const double w = 0.5 + 0.5 * cos(2. * M_PI * (i - F / 2) / F);     // fft_window[i], quisk.c:5091-5096
        sam[2 * (size_t)i] *= w; sam[2 * (size_t)i + 1] *= w;
    }
    int rc = fft_any_host(e, sam.data(), W, 0, X.data());
    if (rc != QK_OK) return rc;
    const double scale = 1. / W;
    int i = 0;
    auto db = [&](int k) {
        const double a = hypot(X[2 * (size_t)k], X[2 * (size_t)k + 1]) * scale;
        return 20.0 * (a <= 1e-7 ? -7.0 : log10(a));
    };
    for (int k = W / 2; k < W; k++, i++) out[i] = db(k);
    for (int k = 0; k < W / 2; k++, i++) out[i] = db(k);
    return QK_OK;
}

int qk_get_filter_rate(int sample_rate, int mode, int bandwidth) { return filter_rate(sample_rate, mode, bandwidth); }
int qk_plan_decimation(int sample_rate, int *d2, int *d3, int *d5) { return plan_decimation(sample_rate, d2, d3, d5); }
int qk_decim_rate(int sample_rate) { return decim_rate(sample_rate); }

int qk_get_stats(qk_engine *e, qk_stats *out)
{
    ENGINE_ENTER(e);
    if (!out) return fail(QK_EINVAL, "null argument");
    int rc = drain_events(e);
    if (rc != QK_OK) return rc;
    *out = e->stats;
    return QK_OK;
}

int qk_set_timing(qk_engine *e, int on)
{
    ENGINE_ENTER(e);
    int rc = drain_events(e);
    e->timing = on == 1 ? 0x1ffu : (unsigned)on >> 1;
     // 1 = everything; otherwise bit (b + 1) selects group b
    return rc;
}

int qk_reset_stats(qk_engine *e)
{
    ENGINE_ENTER(e);
    int rc = drain_events(e);
    e->stats = qk_stats{};
    return rc;
}

int qk_export_audio_device(qk_engine *e, double *dst_dev, size_t dst_stride)
{
    ENGINE_ENTER(e);
    HIPCHECK(hipStreamSynchronize(e->agc_stream));       // the AGC pass writes the audio output
    if (e->apply_stream) HIPCHECK(hipStreamSynchronize(e->apply_stream));
    if (!dst_dev) return fail(QK_EINVAL, "null destination");
    size_t w = 0;
    for (int r = 0; r < e->nrx; r++) w = std::max(w, (size_t)e->out_count[r]);
    if (w > dst_stride) return fail(QK_ERANGE, "stride smaller than the audio block");
    if (w > 0)
        HIPCHECK(hipMemcpy2DAsync(dst_dev, dst_stride * sizeof(double2), e->d_out, e->out_stride * sizeof(double2),
                                  w * sizeof(double2), e->nrx, e->small_mode ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, e->stream));
    SYNC_ALL(e);
    AGC_ABORT_CHECK(e);
    return QK_OK;
}

int qk_debug_tap(qk_engine *e, int which, int rx, double *out, int cap)
{
    ENGINE_ENTER(e); RX_CHECK(e, rx);
    if (which == QK_TAP_ROUTE) {        // the pair of a live stereo route in front of its AGCs: out[0] = left, out[1] = right
        for (auto &r : e->routes) {
            if (r.rx != rx || !r.live) continue;
            SYNC_ALL(e);
            const Stream &p = r.tail.streams[r.tail.play];
            const int m = std::min(p.n, cap);
            for (int ch = 0; ch < 2 && m > 0; ch++)
                HIPCHECK(hipMemcpy(out + (size_t)ch * cap, p.d + (long long)ch * p.rx_stride() + p.hist, sizeof(double) * m, hipMemcpyDeviceToHost));
            return m;
        }
        return fail(QK_EINVAL, "no live route on this receiver");
    }
    const Stream *s = nullptr;
    Chain &c = e->cls_chain[e->cls[rx]];
    switch (which) {
    case QK_TAP_FRONTEND: s = &e->front.streams[0]; break;
    case QK_TAP_DECIM: s = &e->front.streams[e->front.aud]; break;
    case QK_TAP_DEMOD:
        for (auto &st : c.stages) if (st.kind == ST_RXF) s = &c.streams[st.out];
        break;
    case QK_TAP_AUDIO48: if (c.built) s = &c.streams[c.aud]; break;
    case QK_TAP_PLAY: if (c.built) s = &c.streams[c.play]; break;
    }
    if (!s || !s->d) return fail(QK_EINVAL, "no such tap");
    SYNC_ALL(e);
    const int m = std::min(s->n, cap);
    if (s->f32) {      // a float stream: widened on the host
        std::vector<float> tmp((size_t)std::max(m, 1));
        for (int p = 0; p < s->planes && m > 0; p++) {
            HIPCHECK(hipMemcpy(tmp.data(), reinterpret_cast<const float *>(s->d) + (long long)rx * s->rx_stride() + (long long)p * (s->hist + s->cap) + s->hist,
                               sizeof(float) * m, hipMemcpyDeviceToHost));
            for (int i = 0; i < m; i++) out[(size_t)p * cap + i] = (double)tmp[i];
        }
        return m;
    }
    for (int p = 0; p < s->planes && m > 0; p++)
        HIPCHECK(hipMemcpy(out + (size_t)p * cap, s->d + (long long)rx * s->rx_stride() + (long long)p * (s->hist + s->cap) + s->hist,
                           sizeof(double) * m, hipMemcpyDeviceToHost));
    return m;
}

// ---- filter.c drop-ins over host buffers ----
// A grow-only workspace per device, shared by every drop-in call (they take its lock): device input / output buffers and two
// pinned staging buffers, so that a call costs one host-to-device copy (history, block and taps together), its kernel and one
// device-to-host copy -- no allocation, no pageable-memory copies.  Held until the process ends.
namespace {
struct DropWS { unsigned char *d_in = nullptr, *d_out = nullptr, *h_in = nullptr, *h_out = nullptr; size_t c_in = 0, c_out = 0; hipStream_t st = nullptr; };
std::mutex g_ws_mu;
DropWS g_ws[64];
}
static int ws_get(DropWS **out, size_t need_in, size_t need_out)
{
    int ndev = 0, dev = -1;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(QK_EHIP, "no HIP device: the engine has no CPU path");
    HIPCHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(QK_ERANGE, "device ordinal beyond the workspace table");
    DropWS &w = g_ws[dev];
    if (!w.st) HIPCHECK(hipStreamCreateWithFlags(&w.st, hipStreamNonBlocking));
    auto grow = [&](unsigned char *&d, unsigned char *&h, size_t &cap, size_t need) -> int {
        if (need <= cap) return QK_OK;
        if (d) hipFree(d);
        if (h) hipHostFree(h);
        d = nullptr; h = nullptr; cap = 0;
        const size_t want = need + need / 2 + 4096;
        HIPCHECK(hipMalloc(&d, want));
        HIPCHECK(hipHostMalloc(&h, want, hipHostMallocDefault));
        cap = want;
        return QK_OK;
    };
    int rc = grow(w.d_in, w.h_in, w.c_in, need_in);
    if (rc == QK_OK) rc = grow(w.d_out, w.h_out, w.c_out, need_out);
    if (rc != QK_OK) return rc;
    *out = &w;
    return QK_OK;
}
// the decimating FIR over the staged block: the tiled kernel where its tile fits 48 KiB of LDS, else the plain one
static void launch_cfir(hipStream_t st, const double2 *x, double2 *y, const double *h, int T, int D, int first, int n_out, int cpx)
{
    const size_t lds2 = 16 * ((size_t)D * 511 + T), lds1 = 16 * ((size_t)D * 255 + T);
    if (lds2 <= 48 * 1024) hipLaunchKernelGGL(k_cfir_lds<2>, dim3((n_out + 511) / 512), dim3(256), lds2, st, x, y, h, T, D, first, n_out, cpx);
    else if (lds1 <= 48 * 1024) hipLaunchKernelGGL(k_cfir_lds<1>, dim3((n_out + 255) / 256), dim3(256), lds1, st, x, y, h, T, D, first, n_out, cpx);
    else hipLaunchKernelGGL(k_cfir_interleaved, dim3((n_out + 255) / 256), dim3(256), 0, st, x, y, h, T, D, first, n_out, cpx);
}

static int fir_dropin(double *samples, int count, const double *coefs, int ntaps, int decim, double *hist, int *phase, int cpx)
{
    if (!samples || !coefs || !hist || !phase || count < 0 || ntaps < 1 || decim < 1 || *phase < 0 || *phase >= decim)
        return fail(QK_EINVAL, "bad FIR arguments");
    const int n_out = (*phase + count) / decim, first = decim - 1 - *phase;
    const size_t nh = (size_t)ntaps - 1, ntot = nh + (size_t)count, ncoef = (size_t)ntaps * (cpx ? 2 : 1);
    std::lock_guard<std::mutex> lk(g_ws_mu);
    DropWS *w = nullptr;
    int rc = ws_get(&w, ntot * sizeof(double2) + ncoef * sizeof(double), std::max<size_t>(n_out, 1) * sizeof(double2));
    if (rc != QK_OK) return rc;
    // history, block and taps go over in one copy
    if (nh) memcpy(w->h_in, hist, nh * sizeof(double2));
    if (count) memcpy(w->h_in + nh * sizeof(double2), samples, (size_t)count * sizeof(double2));
    memcpy(w->h_in + ntot * sizeof(double2), coefs, ncoef * sizeof(double));
    HIPCHECK(hipMemcpyAsync(w->d_in, w->h_in, ntot * sizeof(double2) + ncoef * sizeof(double), hipMemcpyHostToDevice, w->st));
    if (n_out > 0) {
        launch_cfir(w->st, reinterpret_cast<const double2 *>(w->d_in), reinterpret_cast<double2 *>(w->d_out),
                    reinterpret_cast<const double *>(w->d_in + ntot * sizeof(double2)), ntaps, decim, first, n_out, cpx);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipMemcpyAsync(w->h_out, w->d_out, (size_t)n_out * sizeof(double2), hipMemcpyDeviceToHost, w->st));
    }
    HIPCHECK(hipStreamSynchronize(w->st));
    // new history = the last ntaps-1 samples of (old history + block): the host has them already
    if (nh) memcpy(hist, w->h_in + (ntot - nh) * sizeof(double2), nh * sizeof(double2));
    if (n_out > 0) memcpy(samples, w->h_out, (size_t)n_out * sizeof(double2));
    *phase = (*phase + count) % decim;
    return n_out;
}

// quisk_cInterpolate / quisk_dC_out / quisk_dD_out over host buffers with explicit state
static int hist_dropin(const void *in, size_t in_elem, int count, const double *coefs, size_t coef_doubles, int ntaps, void *hist,
                       void *out, size_t out_elem, size_t n_out, int kind, int L)
{
    const size_t nh = (size_t)ntaps - 1, ntot = nh + (size_t)count;
    std::lock_guard<std::mutex> lk(g_ws_mu);
    DropWS *w = nullptr;
    const size_t in_bytes = (ntot * in_elem + 15) / 16 * 16;
    int rc = ws_get(&w, in_bytes + coef_doubles * sizeof(double), std::max<size_t>(n_out, 1) * out_elem);
    if (rc != QK_OK) return rc;
    if (nh) memcpy(w->h_in, hist, nh * in_elem);
    if (count) memcpy(w->h_in + nh * in_elem, in, (size_t)count * in_elem);
    memcpy(w->h_in + in_bytes, coefs, coef_doubles * sizeof(double));
    HIPCHECK(hipMemcpyAsync(w->d_in, w->h_in, in_bytes + coef_doubles * sizeof(double), hipMemcpyHostToDevice, w->st));
    if (n_out > 0) {
        const dim3 grid((unsigned)((n_out + 255) / 256));
        const double *dh = reinterpret_cast<const double *>(w->d_in + in_bytes);
        if (kind == 0) hipLaunchKernelGGL(k_cinterp_interleaved, grid, dim3(256), 0, w->st, (const double2 *)w->d_in, (double2 *)w->d_out, dh, ntaps, L, count);
        else hipLaunchKernelGGL(k_dfir_out, grid, dim3(256), 0, w->st, (const double *)w->d_in, (double *)w->d_out, dh, ntaps, count, kind == 2);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipMemcpyAsync(w->h_out, w->d_out, n_out * out_elem, hipMemcpyDeviceToHost, w->st));
    }
    HIPCHECK(hipStreamSynchronize(w->st));
    if (nh) memcpy(hist, w->h_in + (ntot - nh) * in_elem, nh * in_elem);
    if (n_out > 0) memcpy(out, w->h_out, n_out * out_elem);
    return (int)n_out;
}

int qk_cInterpolate(double *samples, int count, const double *coefs, int ntaps, int interp, double *hist)
{
    if (!samples || !coefs || !hist || count < 0 || ntaps < 1 || interp < 1) return fail(QK_EINVAL, "bad interpolator arguments");      // (nTaps / interp terms per phase: a remainder is left out, filter.c:130)
    // the reference stops writing at SAMP_BUFFER_SIZE * 8 / 10 outputs (filter.c:135): a caller's buffer sized like the reference's is
    // never overrun; the history still takes in every input sample
    const size_t n_out = std::min((size_t)count * interp, (size_t)66000 * 8 / 10);
    return hist_dropin(samples, sizeof(double2), count, coefs, (size_t)ntaps, ntaps, hist, samples, sizeof(double2), n_out, 0, interp);
}
int qk_dC_out_block(const double *samples, int count, const double *cpx_coefs, int ntaps, double *hist, double *out_complex)
{
    if (!samples || !cpx_coefs || !hist || !out_complex || count < 0 || ntaps < 1) return fail(QK_EINVAL, "bad filter arguments");
    return hist_dropin(samples, sizeof(double), count, cpx_coefs, (size_t)2 * ntaps, ntaps, hist, out_complex, sizeof(double2), (size_t)count, 2, 1);
}
int qk_dD_out_block(double *samples, int count, const double *coefs, int ntaps, double *hist)
{
    if (!samples || !coefs || !hist || count < 0 || ntaps < 1) return fail(QK_EINVAL, "bad filter arguments");
    return hist_dropin(samples, sizeof(double), count, coefs, (size_t)ntaps, ntaps, hist, samples, sizeof(double), (size_t)count, 1, 1);
}

int qk_process_hermes(qk_engine *e, const void *packets, int n_packets, int multirx_count, double *sub_out_host, int *read_errors, int *overrange)
{
    ENGINE_ENTER(e);
    if (!packets || n_packets < 0 || multirx_count < 0 || multirx_count > 9) return fail(QK_EINVAL, "bad Hermes packet arguments");
    const int nrx = multirx_count + 1;
    const int num_records = 504 / (nrx * 6 + 2);                       // quisk.c:3367
    const long long n = (long long)n_packets * 2 * num_records;
    if (n > e->cfg.max_block) return fail(QK_ERANGE, "packets hold more samples than max_block");
    // the bytes the host has to look at itself (quisk.c:3403, 3440-3460, 3496-3499): header, sync, the ADC overrange bit of C1
    int errs = 0, over = 0;
    const unsigned char *pk = static_cast<const unsigned char *>(packets);
    for (int p = 0; p < n_packets; p++) {
        const unsigned char *b = pk + (size_t)p * 1032;
        if (b[0] != 0xEF || b[1] != 0xFE || b[2] != 0x01 || b[3] != 0x06) return fail(QK_EINVAL, "not a Hermes receive packet (EF FE 01 06)");
        for (int start = 11; start < 1000; start += 512) {
            if (b[start - 3] != 0x7F || b[start - 2] != 0x7F || b[start - 1] != 0x7F) errs++;
            int dindex = b[start] >> 1;
            if (!(dindex & 0x40)) dindex >>= 2;
            if (dindex == 0 && (b[start + 1] & 0x01)) over++;
        }
    }
    if (read_errors) *read_errors = errs;
    if (overrange) *overrange = over;
    if (n == 0) return process_device(e, e->d_in, 0) == QK_OK ? 0 : QK_EHIP;
    SYNC_ALL(e);
    const size_t bytes = (size_t)n_packets * 1032;
    if (bytes > e->raw_cap) {
        if (e->d_raw) hipFree(e->d_raw);
        e->d_raw = nullptr; e->raw_cap = 0;
        const size_t want = std::max(bytes, (size_t)e->cfg.max_block * 8);
        HIPCHECK(hipMalloc(&e->d_raw, want));
        e->raw_cap = want;
    }
    double2 *d_sub = nullptr;
    if (sub_out_host && multirx_count > 0) HIPCHECK(hipMalloc(&d_sub, sizeof(double2) * (size_t)n * multirx_count));
    {
        hipError_t ce = hipMemcpyAsync(e->d_raw, packets, bytes, hipMemcpyHostToDevice, e->fe_stream);
        if (ce != hipSuccess) { if (d_sub) hipFree(d_sub); return fail(QK_EHIP, hipGetErrorString(ce)); }
    }
    hipLaunchKernelGGL(k_unpack_hermes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->fe_stream, e->d_raw, n_packets, nrx, num_records,
                       e->d_in, d_sub, n);
    if (d_sub) {
        hipError_t ce = hipMemcpyAsync(sub_out_host, d_sub, sizeof(double2) * (size_t)n * multirx_count, hipMemcpyDeviceToHost, e->fe_stream);
        if (ce == hipSuccess) ce = hipStreamSynchronize(e->fe_stream);
        hipFree(d_sub);
        if (ce != hipSuccess) return fail(QK_EHIP, hipGetErrorString(ce));
    }
    int rc = process_device(e, e->d_in, (int)n);
    return rc == QK_OK ? (int)n : rc;
}

// ---- the entry point that allocates nothing (include/qk_engine.h) ----
// Staging for the in-place filter lives in a file-scope device array; each GPU holds its own copy at its own address,
// looked up once per device (hipGetSymbolAddress) and used through that pointer only -- the array's name is a host-side
// stand-in, not a device address.
static __device__ __align__(256) double2 qk_na_scratch[QK_NOALLOC_MAX_SAMPLES];
static std::mutex g_na_mu;
static double2 *g_na_ptr[64] = {nullptr};
static int noalloc_scratch(double2 **out, hipStream_t st)
{
    // the array of the device that will run the kernels: the stream's own device; the current one for the NULL stream
    int dev = -1, cur = -1;
    HIPCHECK(hipGetDevice(&cur));
    dev = cur;
    if (st) HIPCHECK(hipStreamGetDevice(st, &dev));
    if (dev != cur) return fail(QK_EINVAL, "quisk_cCplxSampleFilter_noalloc: the stream belongs to another device than the current one");
    if (dev < 0 || dev >= 64) return fail(QK_ERANGE, "device ordinal beyond the scratch table");
    std::lock_guard<std::mutex> lk(g_na_mu);
    if (!g_na_ptr[dev]) {
        double2 *p = nullptr;
        HIPCHECK(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(qk_na_scratch)));
        if (!p) return fail(QK_EHIP, "hipGetSymbolAddress returned a null scratch address");
        g_na_ptr[dev] = p;
    }
    *out = g_na_ptr[dev];
    return QK_OK;
}

int quisk_cCplxSampleFilter_noalloc(double *samples_dev, int count, const double *coefs_dev, int ntaps, int decim,
                                    double *hist_dev, int *phase, void *hip_stream)
{
    if (!samples_dev || !coefs_dev || !hist_dev || !phase || count < 0 || ntaps < 1 || decim < 1 || *phase < 0 || *phase >= decim)
        return fail(QK_EINVAL, "bad FIR arguments");
    const int n_out = (*phase + count) / decim, first = decim - 1 - *phase;
    const long long nh = (long long)ntaps - 1, ntot = nh + count;
    if (ntot + n_out > (long long)QK_NOALLOC_MAX_SAMPLES)
        return fail(QK_ERANGE, "quisk_cCplxSampleFilter_noalloc: (ntaps - 1 + count) + outputs exceeds the fixed scratch of "
                               "QK_NOALLOC_MAX_SAMPLES complex samples; split the block");
    double2 *scratch = nullptr;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    int rc = noalloc_scratch(&scratch, st);
    if (rc != QK_OK) return rc;
    double2 *cat = scratch, *y = scratch + ntot;
    if (ntot > 0)
        hipLaunchKernelGGL(k_na_gather, dim3((unsigned)((ntot + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const double2 *>(hist_dev),
                           reinterpret_cast<const double2 *>(samples_dev), cat, (int)nh, count);
    if (n_out > 0) launch_cfir(st, cat, y, coefs_dev, ntaps, decim, first, n_out, 0);
    const long long m = std::max<long long>(n_out, nh);
    if (m > 0)
        hipLaunchKernelGGL(k_na_scatter, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, cat, y, reinterpret_cast<double2 *>(samples_dev),
                           reinterpret_cast<double2 *>(hist_dev), (int)nh, count, n_out);
    HIPCHECK(hipGetLastError());
    *phase = (*phase + count) % decim;
    return n_out;
}

// ---- multi-GPU: the gather of the outputs over RCCL (xGMI), from C ----
static int rccl_load()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return QK_OK;
    // The copy that belongs to the HIP runtime this library is bound to: a process may hold a second ROCm stack (PyTorch wheels
    // bundle one, same sonames), and an RCCL of the other stack would bring up an HSA runtime of its own and find no device.
    // So: the directory of the libamdhip64 that serves our calls first, by path; its own symbols first (RTLD_DEEPBIND).
    void *h = nullptr;
    Dl_info info;
    if (dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
        std::string dir(info.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) {
            dir.resize(slash + 1);
            for (const char *nm : {"librccl.so.1", "librccl.so"}) {
                h = dlopen((dir + nm).c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
                if (h) break;
            }
        }
    }
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
    if (!h) return fail(QK_EUNSUPPORTED, std::string("librccl not found: ") + dlerror());
#define QK_SYM(field, name) do { *(void **)(&g_rccl.field) = dlsym(h, name); if (!g_rccl.field) return fail(QK_EUNSUPPORTED, "librccl lacks " name); } while (0)
    QK_SYM(GetUniqueId, "ncclGetUniqueId"); QK_SYM(CommInitRank, "ncclCommInitRank"); QK_SYM(CommDestroy, "ncclCommDestroy");
    QK_SYM(Send, "ncclSend"); QK_SYM(Recv, "ncclRecv"); QK_SYM(GroupStart, "ncclGroupStart"); QK_SYM(GroupEnd, "ncclGroupEnd");
    QK_SYM(GetErrorString, "ncclGetErrorString");
#undef QK_SYM
    g_rccl.lib = h;
    return QK_OK;
}
#define NCCLCHECK(expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) return fail(QK_EHIP, std::string(#expr) + ": " + g_rccl.GetErrorString(_r)); } while (0)

int qk_comm_unique_id(void *id_out)
{
    if (!id_out) return fail(QK_EINVAL, "null argument");
    int rc = rccl_load();
    if (rc != QK_OK) return rc;
    ncclUniqueId id;
    NCCLCHECK(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return QK_OK;
}

int qk_comm_init(qk_engine *e, const void *id, int rank, int world)
{
    ENGINE_ENTER(e);
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(QK_EINVAL, "bad communicator arguments");
    if (e->comm) return fail(QK_EINVAL, "the engine already has a communicator");
    int rc = rccl_load();
    if (rc != QK_OK) return rc;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    NCCLCHECK(g_rccl.CommInitRank(&e->comm, world, uid, rank));
    e->comm_rank = rank; e->comm_world = world;
    if (e->nrx > 0) HIPCHECK(hipMalloc(&e->d_counts, sizeof(int) * e->nrx));
    return QK_OK;
}

int qk_comm_destroy(qk_engine *e)
{
    ENGINE_ENTER(e);
    if (e->comm) { SYNC_ALL(e); g_rccl.CommDestroy(e->comm); e->comm = nullptr; }
    if (e->d_counts) { hipFree(e->d_counts); e->d_counts = nullptr; }
    return QK_OK;
}

int qk_gather_audio(qk_engine *e, int dst_rank, double *dst_dev, int *counts_dev)
{
    ENGINE_ENTER(e);
    if (!e->comm) return fail(QK_EINVAL, "no communicator (qk_comm_init)");
    if (dst_rank < 0 || dst_rank >= e->comm_world || e->nrx <= 0) return fail(QK_EINVAL, "bad gather arguments");
    const bool root = e->comm_rank == dst_rank;
    if (root && (!dst_dev || !counts_dev)) return fail(QK_EINVAL, "the destination rank needs its buffers");
    // the AGC passes write the audio rows; the sends come behind them on the chain's stream
    HIPCHECK(hipStreamSynchronize(e->agc_stream));
    if (e->apply_stream) HIPCHECK(hipStreamSynchronize(e->apply_stream));
    AGC_ABORT_CHECK(e);
    HIPCHECK(hipMemcpyAsync(e->d_counts, e->out_count.data(), sizeof(int) * e->nrx, hipMemcpyHostToDevice, e->stream));
    const size_t row = (size_t)e->nrx * e->out_stride * 2;          // doubles per rank
    NCCLCHECK(g_rccl.GroupStart());
    NCCLCHECK(g_rccl.Send(e->d_out, row, ncclDouble, dst_rank, e->comm, e->stream));
    NCCLCHECK(g_rccl.Send(e->d_counts, (size_t)e->nrx, ncclInt, dst_rank, e->comm, e->stream));
    if (root)
        for (int r = 0; r < e->comm_world; r++) {
            NCCLCHECK(g_rccl.Recv(dst_dev + (size_t)r * row, row, ncclDouble, r, e->comm, e->stream));
            NCCLCHECK(g_rccl.Recv(counts_dev + (size_t)r * e->nrx, (size_t)e->nrx, ncclInt, r, e->comm, e->stream));
        }
    NCCLCHECK(g_rccl.GroupEnd());
    HIPCHECK(hipStreamSynchronize(e->stream));
    return QK_OK;
}

int qk_gather_graph(qk_engine *e, int dst_rank, double *dst_dev)
{
    ENGINE_ENTER(e);
    if (!e->comm) return fail(QK_EINVAL, "no communicator (qk_comm_init)");
    if (dst_rank < 0 || dst_rank >= e->comm_world || e->fftN <= 0) return fail(QK_EINVAL, "bad gather arguments");
    const bool root = e->comm_rank == dst_rank;
    if (root && !dst_dev) return fail(QK_EINVAL, "the destination rank needs its buffer");
    const size_t W = (size_t)e->fftW;
    HIPCHECK(hipStreamSynchronize(e->fe_stream));                    // the release kernel writes the row on the front end's stream
    NCCLCHECK(g_rccl.GroupStart());
    NCCLCHECK(g_rccl.Send(e->d_graph, W, ncclDouble, dst_rank, e->comm, e->stream));
    if (root)
        for (int r = 0; r < e->comm_world; r++) NCCLCHECK(g_rccl.Recv(dst_dev + (size_t)r * W, W, ncclDouble, r, e->comm, e->stream));
    NCCLCHECK(g_rccl.GroupEnd());
    HIPCHECK(hipStreamSynchronize(e->stream));
    return QK_OK;
}

long long qk_debug_agc_passes(qk_engine *e, int rx)
{   /* statistics: speculation passes the AGC kernel has run for this receiver (debug) */
    if (!e || rx < 0 || rx >= e->nrx) return -1;
    std::lock_guard<std::mutex> lk(e->mu);
    hipSetDevice(e->cfg.device);
    hipStreamSynchronize(e->fe_stream); hipStreamSynchronize(e->stream); hipStreamSynchronize(e->agc_stream);
    if (e->apply_stream) hipStreamSynchronize(e->apply_stream);
    AgcState st;
    if (hipMemcpy(&st, e->d_agc + rx, sizeof st, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return st.pad;
}

int quisk_cCplxSampleFilter(double *samples, int count, const double *coefs, int ntaps, int decim, double *hist, int *phase)
{ return fir_dropin(samples, count, coefs, ntaps, decim, hist, phase, 0); }
int quisk_ccfFilter(double *samples, int count, const double *cpx_coefs, int ntaps, int decim, double *hist, int *phase)
{ return fir_dropin(samples, count, cpx_coefs, ntaps, decim, hist, phase, 1); }

}  // extern "C"
