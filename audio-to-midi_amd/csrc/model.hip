// Context, flat parameter layout, and forward / backward orchestration behind the C ABI (amt.h).
//
// Reference path reproduced here:
//   OutputSequenceGenerator.__call__   model.py:740-769   (forward)
//   compute_loss / value_and_grad      train.py:39-62     (loss + backward)
//   compute_training_step (grad part)  train.py:292-322   (fp32 accumulation, unscale, finite flag)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/amt.h"
#include "../../include/amt_ops.h"
#include "kernels.h"
#include "prof.h"

using namespace amt;

namespace {

std::string g_create_error;

struct Seg {
  std::string name;
  size_t off;
  int rows, cols;
  int lr_depth;
};

struct BlockP {  // offsets into the flat fp32 vector
  size_t dw_w, dw_b, ln_w, ln_b, pw1_w, pw1_b, pw2_w, pw2_b, gamma;
  size_t c_pw1, c_pw1t, c_pw2, c_pw2t;  // compute-copy offsets (elements of T)
  size_t c_pw2tg;                       // pw2^T with its columns scaled by gamma (layer scale folded)
  size_t c_pw1p, c_pw2p;                // fragment-packed copies for the chain kernels (C >= 16)
  size_t c_pw2tgp, c_pw1tp;             // packed pw2^T * gamma and pw1^T (backward chain)
};
struct StageP {
  int C, Cin, depth;
  size_t conv_w, conv_b, ln_w, ln_b;   // stem / downsample
  size_t c_w, c_wt;                    // downsample compute copies
  std::vector<BlockP> blocks;
};
struct LayerP {
  size_t ln1_w, ln1_b, wqc, wkv, wo, ln2_w, ln2_b, w1, b1, w2, b2;
  size_t c_wqc, c_wqct, c_wkv, c_wkvt, c_wo, c_wot, c_w1, c_w1t, c_w2, c_w2t;
  size_t c_wop, c_w1p, c_w2p, c_wqcp, c_wkvp;   // fragment-packed copies for the chain kernels
  size_t c_wotp, c_wqctp, c_wkvtp;              // packed transposes (backward chains)
  size_t c_w2tp, c_w1tp;                        // packed transposes of the FFN weights (ffn_bwd_fused)
};

struct Layout {
  std::vector<Seg> segs;
  std::vector<StageP> stages;
  std::vector<LayerP> layers;
  size_t fn_w, fn_b;                       // final CNN norm
  size_t hd_lnw, hd_lnb, hd_w, hd_b;       // decoder
  size_t c_hw, c_hwt;                      // head compute copies ([96][H], [H][96])
  size_t n_params = 0;
  size_t n_compute = 0;                    // elements of the compute-copy buffer
  std::vector<ConvDesc> conv;
  std::vector<int> conv_need;              // per desc: which execution mode reads the copy (ConvNeed)
  int H = 0, QC = 0, KVW = 0, I2 = 0, VP = 0;   // hidden, q+c width, k+v width, 2*inter, padded events
};

size_t align8(size_t x) { return (x + 7) & ~(size_t)7; }

// who reads a compute copy: amt_create drops the conversions nobody reads in the selected mode
// (convert_batch runs after every optimizer step: 69 -> ~35 us with the default chains)
enum ConvNeed { NEED_ALWAYS = 0, NEED_UNFUSED_BLOCK, NEED_FUSED_BLOCK, NEED_UNFUSED_TR, NEED_FUSED_TR, NEED_FFN_CHAIN,
                NEED_FFN_TILED };   // row-major W1 / W2 copies: only the tiled-GEMM form of the FFN reads them

bool validate(const amt_config& c, std::string& err) {
  if (c.n_stages < 1 || c.n_stages > AMT_MAX_STAGES) { err = "n_stages out of range"; return false; }
  for (int i = 0; i < c.n_stages; ++i) {
    if (c.dims[i] != (4 << i)) { err = "dims[i] must equal 4 << i"; return false; }
    if (c.depths[i] < 0) { err = "negative depth"; return false; }
  }
  int n_blocks = 0;
  for (int i = 0; i < c.n_stages; ++i) n_blocks += c.depths[i];
  if (n_blocks > 64) { err = "at most 64 CNN Blocks in total (sum of depths; the reference has 39)"; return false; }
  int H = c.dims[c.n_stages - 1];
  if (H != 256) { err = "hidden size (dims[-1]) must be 256"; return false; }
  if (c.n_heads * c.head_dim != H || c.head_dim != 64) { err = "n_heads*head_dim must be 256 with head_dim 64"; return false; }
  if (c.kv_compress != 64) { err = "kv_compress must be 64"; return false; }
  if (c.ffn_inter % 16 != 0 || c.ffn_inter <= 0) { err = "ffn_inter must be a positive multiple of 16"; return false; }
  if (c.n_events <= 0 || c.n_events > 1024) { err = "n_events out of range"; return false; }
  if (c.local_window != 16) { err = "local_window must be 16"; return false; }
  if (c.act_dtype != AMT_F32 && c.act_dtype != AMT_BF16) { err = "act_dtype must be AMT_F32 or AMT_BF16"; return false; }
  if (!(c.dropout_rate >= 0.f && c.dropout_rate < 1.f)) { err = "dropout_rate must be in [0,1)"; return false; }
  if (c.n_layer_pairs < 0) { err = "n_layer_pairs negative"; return false; }
  return true;
}

Layout build_layout(const amt_config& c) {
  Layout L;
  size_t off = 0, coff = 0;
  auto seg = [&](const std::string& name, int rows, int cols, int depth) {
    size_t o = off;
    L.segs.push_back(Seg{name, o, rows, cols, depth});
    off += (size_t)rows * cols;
    return o;
  };
  auto cbuf = [&](size_t n) { size_t o = coff; coff += align8(n); return o; };
  int need_cur = NEED_ALWAYS;   // tag of the descriptors pushed next
  auto push = [&](const ConvDesc& d) { L.conv.push_back(d); L.conv_need.push_back(need_cur); };
  auto cpair = [&](size_t src, int rows, int cols, size_t& cw, size_t& cwt) {
    cw = cbuf((size_t)rows * cols);
    cwt = cbuf((size_t)rows * cols);
    push(ConvDesc{src, cw, rows, cols, rows, 0, -1, 0});
    push(ConvDesc{src, cwt, cols, rows, rows, 1, -1, 0});
  };
  auto cpacked = [&](size_t src, int rows, int cols) {
    size_t o = cbuf(chain_packed_elems(rows, cols));
    push(ConvDesc{src, o, rows, cols, rows, 0, -1, 1});
    return o;
  };
  int depth_base = 0;
  for (int s = 0; s < c.n_stages; ++s) {
    StageP st;
    st.C = c.dims[s];
    st.Cin = s == 0 ? 2 : c.dims[s - 1];
    st.depth = c.depths[s];
    std::string p = "layers." + std::to_string(s) + ".0.";
    int d0 = depth_base;  // Sequential index 0 (train.py:646-672: depth = sum(depths[:s]) + idx)
    if (s == 0) {
      st.conv_w = seg(p + "conv.weight", st.C, 10, d0);
      st.conv_b = seg(p + "conv.bias", st.C, 1, d0);
      st.ln_w = seg(p + "norm.weight", st.C, 1, d0);
      st.ln_b = seg(p + "norm.bias", st.C, 1, d0);
      st.c_w = st.c_wt = 0;
    } else {
      st.ln_w = seg(p + "norm.weight", st.Cin, 1, d0);
      st.ln_b = seg(p + "norm.bias", st.Cin, 1, d0);
      st.conv_w = seg(p + "conv.weight", st.C, 2 * st.Cin, d0);
      st.conv_b = seg(p + "conv.bias", st.C, 1, d0);
      cpair(st.conv_w, st.C, 2 * st.Cin, st.c_w, st.c_wt);
    }
    for (int j = 1; j <= st.depth; ++j) {
      BlockP b;
      std::string q = "layers." + std::to_string(s) + "." + std::to_string(j) + ".";
      int d = depth_base + j;
      int C = st.C;
      b.dw_w = seg(q + "depth_conv.weight", 7, C, d);
      b.dw_b = seg(q + "depth_conv.bias", C, 1, d);
      b.ln_w = seg(q + "norm.weight", C, 1, d);
      b.ln_b = seg(q + "norm.bias", C, 1, d);
      b.pw1_w = seg(q + "point_conv_1.weight", 2 * C, C, d);
      b.pw1_b = seg(q + "point_conv_1.bias", 2 * C, 1, d);
      b.pw2_w = seg(q + "point_conv_2.weight", C, 2 * C, d);
      b.pw2_b = seg(q + "point_conv_2.bias", C, 1, d);
      b.gamma = seg(q + "gamma", C, 1, d);
      need_cur = C >= 16 ? NEED_UNFUSED_BLOCK : NEED_ALWAYS;   // C <= 8: the per-position kernels read these
      cpair(b.pw1_w, 2 * C, C, b.c_pw1, b.c_pw1t);
      cpair(b.pw2_w, C, 2 * C, b.c_pw2, b.c_pw2t);
      b.c_pw2tg = cbuf((size_t)2 * C * C);
      push(ConvDesc{b.pw2_w, b.c_pw2tg, 2 * C, C, C, 1, (long long)b.gamma, 0});
      need_cur = NEED_FUSED_BLOCK;
      b.c_pw1p = b.c_pw2p = b.c_pw2tgp = b.c_pw1tp = 0;
      if (C >= 16) {
        b.c_pw1p = cpacked(b.pw1_w, 2 * C, C);
        b.c_pw2p = cpacked(b.pw2_w, C, 2 * C);
        b.c_pw2tgp = cbuf(chain_packed_elems(2 * C, C));
        push(ConvDesc{b.pw2_w, b.c_pw2tgp, 2 * C, C, C, 1, (long long)b.gamma, 1});
        b.c_pw1tp = cbuf(chain_packed_elems(C, 2 * C));
        push(ConvDesc{b.pw1_w, b.c_pw1tp, C, 2 * C, 2 * C, 1, -1, 1});
      }
      need_cur = NEED_ALWAYS;
      st.blocks.push_back(b);
    }
    depth_base += st.depth;
    L.stages.push_back(st);
  }
  const int H = c.dims[c.n_stages - 1];
  const int HD = c.n_heads * c.head_dim;
  L.H = H;
  L.QC = HD + c.kv_compress;
  L.KVW = 2 * HD;
  L.I2 = 2 * c.ffn_inter;
  L.VP = (c.n_events + 31) / 32 * 32;
  L.fn_w = seg("norm.weight", H, 1, -1);
  L.fn_b = seg("norm.bias", H, 1, -1);
  for (int li = 0; li < 2 * c.n_layer_pairs; ++li) {
    LayerP l;
    std::string t = "tl." + std::to_string(li) + ".";
    l.ln1_w = seg(t + "ln1.weight", H, 1, -1);
    l.ln1_b = seg(t + "ln1.bias", H, 1, -1);
    l.wqc = seg(t + "wqc", L.QC, H, -1);
    l.wkv = seg(t + "wkv", L.KVW, c.kv_compress, -1);
    l.wo = seg(t + "wo", H, HD, -1);
    l.ln2_w = seg(t + "ln2.weight", H, 1, -1);
    l.ln2_b = seg(t + "ln2.bias", H, 1, -1);
    l.w1 = seg(t + "w1", L.I2, H, -1);
    l.b1 = seg(t + "b1", L.I2, 1, -1);
    l.w2 = seg(t + "w2", H, c.ffn_inter, -1);
    l.b2 = seg(t + "b2", H, 1, -1);
    need_cur = NEED_UNFUSED_TR;
    cpair(l.wqc, L.QC, H, l.c_wqc, l.c_wqct);
    cpair(l.wkv, L.KVW, c.kv_compress, l.c_wkv, l.c_wkvt);
    cpair(l.wo, H, HD, l.c_wo, l.c_wot);
    need_cur = NEED_FFN_TILED;
    cpair(l.w1, L.I2, H, l.c_w1, l.c_w1t);
    cpair(l.w2, H, c.ffn_inter, l.c_w2, l.c_w2t);
    need_cur = NEED_FUSED_TR;
    l.c_wop = cpacked(l.wo, H, HD);
    l.c_wqcp = cpacked(l.wqc, L.QC, H);
    l.c_wkvp = cpacked(l.wkv, L.KVW, c.kv_compress);
    auto cpacked_t = [&](size_t src, int rows, int cols) {   // packed copy of src^T: dst [rows][cols], src [cols][rows]
      size_t o = cbuf(chain_packed_elems(rows, cols));
      push(ConvDesc{src, o, rows, cols, cols, 1, -1, 1});
      return o;
    };
    l.c_wotp = cpacked_t(l.wo, HD, H);
    l.c_wqctp = cpacked_t(l.wqc, H, L.QC);
    l.c_wkvtp = cpacked_t(l.wkv, c.kv_compress, L.KVW);
    need_cur = NEED_FFN_CHAIN;
    l.c_w1p = cpacked(l.w1, L.I2, H);
    l.c_w2p = cpacked(l.w2, H, c.ffn_inter);
    l.c_w2tp = cpacked_t(l.w2, c.ffn_inter, H);
    l.c_w1tp = cpacked_t(l.w1, H, L.I2);
    need_cur = NEED_ALWAYS;
    L.layers.push_back(l);
  }
  L.hd_lnw = seg("decoder.norm.weight", H, 1, -1);
  L.hd_lnb = seg("decoder.norm.bias", H, 1, -1);
  L.hd_w = seg("decoder.decoder_pooling.weight", c.n_events, H, -1);
  L.hd_b = seg("decoder.decoder_pooling.bias", c.n_events, 1, -1);
  // head: padded to VP rows so that every GEMM has M % 4 == 0 and K % 8 == 0
  L.c_hw = cbuf((size_t)L.VP * H);
  L.c_hwt = cbuf((size_t)L.VP * H);
  need_cur = NEED_ALWAYS;
  push(ConvDesc{L.hd_w, L.c_hw, L.VP, H, c.n_events, 0, -1, 0});
  push(ConvDesc{L.hd_w, L.c_hwt, H, L.VP, c.n_events, 1, -1, 0});
  L.n_params = off;
  L.n_compute = coff;
  return L;
}

struct Arena {
  char* base = nullptr;
  size_t cap = 0, off = 0, peak = 0;
  bool dry = false;
  bool overflow = false;
  bool* stop = nullptr;   // also set on overflow: the "dry" switch of the launchers that are called outside RUN()
  void* alloc(size_t bytes) {
    size_t o = (off + 255) & ~(size_t)255;
    off = o + bytes;
    if (off > peak) peak = off;
    if (dry) return reinterpret_cast<void*>(o + 4096);  // fake non-null address, never dereferenced
    if (off > cap) {      // never hand out memory past the plan; every launch after this point is suppressed
      overflow = true;
      if (stop) *stop = true;
      return base;
    }
    return base + o;
  }
};

struct BlockSave { void *x, *ln, *u, *h, *z; float *mean, *rstd; };
struct StageSave { void *x, *ln; float *mean, *rstd; std::vector<BlockSave> blocks; };
struct LayerSave { void *x, *xn, *qc, *kv, *ao, *h, *hn, *u, *g; float *m1, *r1, *m2, *r2, *lse; unsigned long long* keepbits; int prerot; };

}  // namespace

struct amt_ctx {
  amt_config cfg;
  Layout L;
  DType dt;
  int device;
  size_t esz;
  std::string err;
  float* params = nullptr;      // fp32 master [n_params]
  void* wc = nullptr;           // compute copies (T)
  float* head_bias = nullptr;   // [VP] fp32 (padded)
  ConvDesc* conv_dev = nullptr;
  int* conv_prefix_dev = nullptr;
  int conv_tiles = 0;
  float *rope_cos = nullptr, *rope_sin = nullptr;
  int rope_pos = 0;
  bool rope_user = false;
  Arena arena;
  int plan_B = 0, plan_T = 0, plan_train = -1;
  // Every (B, T, mode) planned so far keeps its record descriptors on the device, and the arena / record workspace only
  // ever grow: switching back to a shape that was planned before reallocates nothing, so a hipGraph captured for it
  // stays valid.  plan_gen counts the events that DO move device memory a captured graph may point into.
  struct PlanRec {
    int B, T, train;
    size_t peak, scratch_max, part_total, ev_needed;
    PartDesc* desc_dev;
    int n_desc, n_half;
  };
  std::vector<PlanRec> plans;
  uint64_t plan_gen = 0;
  // optimizer state
  float *adam_m = nullptr, *adam_v = nullptr, *adam_upd = nullptr, *lr_scale = nullptr, *norm2 = nullptr;
  unsigned char* lr_idx = nullptr;   // lr_scale[i] == lr_tab[lr_idx[i]] (one-launch fused optimizer, head.hip)
  float* lr_tab = nullptr;           // [256]
  int* persist_fail = nullptr;       // mapped host word the one-launch optimizer sets when its grid barrier times out
  int* persist_fail_dev = nullptr;   // ... its device address
  float lr_scale_decay = -1.f;
  // saved forward state
  std::vector<StageSave> ssave;
  std::vector<LayerSave> lsave;
  void *fn_x = nullptr, *fn_y = nullptr; float *fn_mean = nullptr, *fn_rstd = nullptr;
  void *hd_x = nullptr, *hd_n = nullptr; float *hd_mean = nullptr, *hd_rstd = nullptr;
  float* logits_pad = nullptr;
  float* loss_acc = nullptr;    // device float[2]
  float* gacc = nullptr;        // internal gradient accumulation buffer [n_params + 64], see run_backward
  int gacc_pad = 0;
  // fused-optimizer protocol (amt_forward_backward* with grads_flat == NULL, then amt_adamw_step_gated with grads_flat ==
  // NULL): the accumulator is handed from the backward pass to the optimizer, which leaves it zeroed
  // Invariant once the protocol has been used (fused_used): BETWEEN steps the accumulator holds zeros - it is zeroed at
  // creation, by the optimizer's commit pass, by amt_grad_accumulator_reset, and at the END of every backward pass that
  // writes the caller's buffer - so a fused backward pass (also a replayed one) never has to zero it first.
  bool fused_used = false;
  float gacc_mul = 1.f;         // 1 / (scale * accum_count) of the sums the last fused backward pass left there
  // per-block partial-sum records of the backward row kernels (kernels.h: PartialSink)
  float* partials = nullptr;
  size_t partials_cap = 0, part_total = 0;
  std::vector<PartDesc> part_descs;
  PartDesc* part_desc_dev = nullptr;
  int n_part_desc = 0;
  int n_part_desc_half = 0;     // records registered by the first half of the backward pass (run_backward phases)
  size_t fwd_end_off = 0;       // arena offset after the forward pass (start of the backward allocations)
  // side stream for weight-gradient launches (they only feed the gradient buffer, so they overlap
  // with the dependent data-gradient chain); events are created at plan time
  hipStream_t side = nullptr;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0, ev_needed = 0;
  size_t scratch_max = 0;       // bytes of per-scope backward scratch (two alternating regions)
  bool use_side = false;
  bool fuse = true;             // chain kernels (chain.hip); AMT_FUSE=0 selects the unfused launch sequence
  int nochain = 0;              // debugging: AMT_NOCHAIN bit mask disables individual chain kernels / riders
  bool attn_fp8 = false;        // amt_set_attn_fp8: global-attention forward on fp8 MFMA (bf16 contexts only)
  mutable bool team_dead = false;   // a team hand-off has timed out (seen by amt_team_status): no further team launches from this context
  bool attn_nobits = false;     // AMT_ATTN_BITS=0: the attention backward re-hashes its dropout masks (round-2 form, for A/B and the equality test)
  float* amax = nullptr;        // [layers][prev q,k,v,- | cur q,k,v,-]: delayed per-tensor scaling state
  uint64_t* seed_word = nullptr;   // device word added to every dropout seed (amt_seed_word / amt_advance_seed)
  TeamState* team = nullptr;    // XCD-team protocol words (team.h); null: no team launches (AMT_TEAM=0 and AMT_TLAYER=0)
  bool labels_f16 = false;      // amt_set_label_dtype: the labels of amt_forward_backward* are IEEE half (grain_loader.py:88)
  bool team_cnn = false;        // AMT_TEAM=1: the Blocks of a CNN stage as one team launch (chain.hip: block_run_fwd)
  bool tlayer = false;          // AMT_TLAYER: the transformer stack's forward as one team launch (tlayer.hip)
  unsigned long long* tl_stamps = nullptr;   // AMT_TLAYER_STAMPS=1: per-phase wall-clock stamps of that launch (amt_tlayer_stamps)
  // run_backward phase 3 ("first half complete" published from inside the launch sequence): the counter, its mapped host
  // copy, and the host's bookkeeping of the value the next wait expects
  GammaDesc* gamma_desc_dev = nullptr;     // one entry per CNN Block (kernels.h: cnn_gamma_finish)
  int n_gamma_desc = 0, gamma_rows = 0;
  unsigned* half_sig = nullptr;            // device counter of published signals
  volatile unsigned* half_host = nullptr;  // the same count in mapped host memory (what amt_first_half_wait polls)
  unsigned* half_host_dev = nullptr;       // ... its device address
  int half_timeouts = 0;
  uint32_t half_base = 0, half_waits = 0;
  int cnn_late_max_c = 0;       // AMT_CNN_WGRAD_LATE=C: weight gradients of the CNN stages with <= C channels as one batch of launches after the pass (1 = all, 0 = none)
  bool cnn_late_env = false;    // not set: none while riders are on, all from 8192 tokens on (run_backward)
  bool fuse_ffn = true;         // AMT_FUSE_FFN=0: second half of a layer as chain + two tiled GEMM launches
  bool lr_idx_ok = false;       // lr_idx / lr_tab describe lr_scale (fewer than 255 depths)
  bool opt_persist = true;      // one-launch fused optimizer (head.hip: adamw_persist_kernel); AMT_OPT_PERSIST=0 or a barrier timeout: two launches
  int n_cus = 0;                // compute units of the device (grid of the one-launch optimizer)
  int persist_grid_extra = 0;   // AMT_OPT_PERSIST_EXTRA=<k> (tests): k workgroups more than can be resident, so that the barrier times out
  unsigned persist_spins = 0;   // AMT_OPT_PERSIST_SPINS=<n> (tests): spin limit of the grid barrier in polls (0: the default, ~1 s)
  int persist_delay_wg = -1;    // AMT_OPT_PERSIST_DELAY=<wg>:<polls> (tests): workgroup <wg> arrives <polls> polls late
  unsigned persist_delay_spins = 0;
  int persist_timeouts = 0;     // launches of the one-launch optimizer that this context has seen time out (amt_optimizer_status)
  bool prerot = true;           // AMT_PREROT=0: global layers' q / k leave the chain raw, the attention kernels rotate them while staging

  const float* P(size_t off) const { return params + off; }
  const void* WC(size_t off) const { return static_cast<const char*>(wc) + off * esz; }
};

namespace {

#define HIPCHK(c, call)                                                                     \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      (c)->err = std::string(#call) + ": " + hipGetErrorString(e_);                         \
      return 1;                                                                             \
    }                                                                                       \
  } while (0)

void build_rope(int hd, int n_pos, std::vector<float>& cs, std::vector<float>& sn) {
  // rope.py:12-22 in float32 arithmetic
  int half = hd / 2;
  cs.resize((size_t)n_pos * half);
  sn.resize((size_t)n_pos * half);
  for (int i = 0; i < half; ++i) {
    float e = (float)(2 * i) / (float)hd;
    float inv = 1.0f / powf(10000.0f, e);
    for (int t = 0; t < n_pos; ++t) {
      float f = (float)t * inv;
      cs[(size_t)t * half + i] = cosf(f);
      sn[(size_t)t * half + i] = sinf(f);
    }
  }
}

int upload_rope(amt_ctx* c, const float* cs, const float* sn, int n_pos) {
  size_t bytes = (size_t)n_pos * (c->cfg.head_dim / 2) * sizeof(float);
  if (c->rope_cos) { (void)hipFree(c->rope_cos); (void)hipFree(c->rope_sin); c->rope_cos = c->rope_sin = nullptr; }
  HIPCHK(c, hipMalloc(&c->rope_cos, bytes));
  HIPCHK(c, hipMalloc(&c->rope_sin, bytes));
  HIPCHK(c, hipMemcpy(c->rope_cos, cs, bytes, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->rope_sin, sn, bytes, hipMemcpyHostToDevice));
  c->rope_pos = n_pos;
  return 0;
}

int ensure_rope(amt_ctx* c, int S) {
  if (c->rope_pos >= S) return 0;
  if (c->rope_user) { c->err = "RoPE table set by amt_set_rope has fewer positions than frames"; return 1; }
  int n = S > 512 ? S : 512;
  std::vector<float> cs, sn;
  build_rope(c->cfg.head_dim, n, cs, sn);
  return upload_rope(c, cs.data(), sn.data(), n);
}

struct Drop { uint32_t thr; float inv_keep; };
Drop drop_params(const amt_ctx* c, bool train) {
  Drop d{0u, 1.f};
  if (train && c->cfg.dropout_rate > 0.f) {
    d.thr = (uint32_t)lrintf(c->cfg.dropout_rate * 16777216.f);
    d.inv_keep = 1.f / (1.f - (float)d.thr / 16777216.f);
  }
  return d;
}
// dropout stream ids: layer li -> attention 2*li, ffn 2*li+1
inline uint32_t stream_attn(int li) { return 2u * (uint32_t)li; }
inline uint32_t stream_ffn(int li) { return 2u * (uint32_t)li + 1u; }

GemmArgs gargs(const void* X, int ldx, const void* W, int ldw, int N, int M, int K, const float* bias, void* Y,
               int ldy) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.X = X; g.ldx = ldx; g.W = W; g.ldw = ldw; g.N = N; g.M = M; g.K = K; g.bias = bias; g.Y = Y; g.ldy = ldy;
  g.inv_keep = 1.f;
  return g;
}

// ------------------------------------------------------------------------------------ forward
// save == true : every activation the backward pass needs stays in the arena.
// save == false: ping-pong buffers; temporaries are recycled per block / layer.
int run_forward(amt_ctx* c, hipStream_t s, const void* audio, int B, int T, bool save, bool train, uint64_t seed) {
  const amt_config& cfg = c->cfg;
  const Layout& L = c->L;
  Arena& A = c->arena;
  const DType dt = c->dt;
  const size_t es = c->esz;
  const bool dry = A.dry;
#define RUN(...) do { if (!dry && !A.overflow) { __VA_ARGS__; } } while (0)   // an overflowed plan launches nothing more
  const Drop dp = drop_params(c, train);

  c->ssave.assign(cfg.n_stages, StageSave());
  c->lsave.assign(L.layers.size(), LayerSave());
  const bool fp8 = c->attn_fp8 && dt == DT_BF16;
  if (fp8) RUN(attn_fp8_roll(s, c->amax, (int)L.layers.size()));

  int Lcur = T / 5;
  const size_t act_elems = (size_t)B * Lcur * 4;   // B * C * L is constant over stages
  void* pp[2] = {nullptr, nullptr};
  if (!save) { pp[0] = A.alloc(act_elems * es); pp[1] = A.alloc(act_elems * es); }
  int ppi = 0;
  const size_t mark0 = A.off;
  auto out_buf = [&](size_t elems) -> void* {
    if (save) return A.alloc(elems * es);
    void* p = pp[ppi]; ppi ^= 1; return p;
  };

  void* cur = nullptr;
  int dbg_slot = 0;   // micro-benchmark builds only (AMT_CHAIN_TIMING)
  // The LayerNorm a Downsample (or the final norm) opens with is evaluated by the launch of the stage's last Block on
  // the y rows it holds (kernels.h: TailLn): one launch less per stage.  AMT_TAIL_LN=0 / debug-mask bit 8192: off.
  static const bool tail_ln_env = !(getenv("AMT_TAIL_LN") && atoi(getenv("AMT_TAIL_LN")) == 0);
  const bool tail_ln = c->fuse && tail_ln_env && !(c->nochain & 8192);
  bool tail_done = false;
  void* fn_y_tail = nullptr;
  for (int si = 0; si < cfg.n_stages; ++si) {
    const StageP& st = L.stages[si];
    StageSave& sv = c->ssave[si];
    const int C = st.C;
    if (si == 0) {
      void* y = out_buf((size_t)B * Lcur * C);
      RUN(stem_fwd(s, dt, audio, c->P(st.conv_w), c->P(st.conv_b), c->P(st.ln_w), c->P(st.ln_b), y, B, T));
      cur = y;
    } else {
      if (!save) A.off = mark0;
      const int Lin = Lcur, Cin = st.Cin;
      Lcur = Lin / 2;
      sv.x = cur;
      if (!tail_done) {
        sv.ln = A.alloc((size_t)B * Lin * Cin * es);
        sv.mean = (float*)A.alloc((size_t)B * Lin * sizeof(float));
        sv.rstd = (float*)A.alloc((size_t)B * Lin * sizeof(float));
        RUN(layernorm_fwd(s, dt, cur, c->P(st.ln_w), c->P(st.ln_b), sv.ln, sv.mean, sv.rstd, B * Lin, Cin));
      }
      tail_done = false;
      void* y = out_buf((size_t)B * Lcur * C);
      GemmArgs g = gargs(sv.ln, 2 * Cin, c->WC(st.c_w), 2 * Cin, B * Lcur, C, 2 * Cin, c->P(st.conv_b), y, C);
      RUN(gemm_nt(s, dt, EPI_STORE_RES, g));   // the stage's residual stream starts here: stored as Res (common.h)
      cur = y;
    }
    const int N = B * Lcur;
    sv.blocks.assign(st.blocks.size(), BlockSave());
    // XCD-team form: the whole stage in one launch (team.h; chain.hip: block_run_fwd)
    if (save && c->fuse && c->team != nullptr && c->team_cnn && !c->team_dead && C >= 16 && st.blocks.size() >= 2 && st.blocks.size() <= (size_t)BLOCK_RUN_MAX &&
        !(c->nochain & 1) && !(c->nochain & 1024)) {
      BlockRunFwdArgs ra;
      memset(&ra, 0, sizeof(ra));
      ra.X0 = cur; ra.st = c->team; ra.B = B; ra.L = Lcur; ra.C = C; ra.n = (int)st.blocks.size(); ra.dbg_base = dbg_slot;
      std::vector<BlockFwdArgs> single(st.blocks.size());
      const void* xin = cur;
      for (size_t j = 0; j < st.blocks.size(); ++j) {
        const BlockP& bp = st.blocks[j];
        BlockSave& bs = sv.blocks[j];
        bs.x = const_cast<void*>(xin);
        bs.ln = A.alloc((size_t)N * C * es);
        bs.mean = (float*)A.alloc((size_t)N * sizeof(float));
        bs.rstd = (float*)A.alloc((size_t)N * sizeof(float));
        bs.u = A.alloc((size_t)N * 2 * C * es);
        bs.h = A.alloc((size_t)N * 2 * C * es);
        bs.z = nullptr;      // not saved: dgamma comes from the raw pw2 weight gradient (kernels.h: cnn_gamma_finish)
        void* y = out_buf((size_t)N * C);
        BlockRunFwdItem& it = ra.it[j];
        it.W1 = c->WC(bp.c_pw1p); it.W2 = c->WC(bp.c_pw2p); it.wdw = c->P(bp.dw_w); it.bdw = c->P(bp.dw_b);
        it.lnw = c->P(bp.ln_w); it.lnb = c->P(bp.ln_b); it.b1 = c->P(bp.pw1_b); it.b2 = c->P(bp.pw2_b); it.gamma = c->P(bp.gamma);
        it.LN = bs.ln; it.U = bs.u; it.Hh = bs.h; it.Z = bs.z; it.Y = y; it.mean = bs.mean; it.rstd = bs.rstd;
        single[j] = BlockFwdArgs{xin, it.wdw, it.bdw, it.lnw, it.lnb, it.W1, it.b1, it.W2, it.b2, it.gamma, bs.ln, bs.mean,
                                 bs.rstd, bs.u, bs.h, bs.z, y, B, Lcur, C, dbg_slot++};
        xin = y;
      }
      cur = const_cast<void*>(xin);
      if (!dry && !A.overflow) {
        if (!block_run_fwd(s, dt, ra))
          for (const BlockFwdArgs& fa : single) block_fwd_fused(s, dt, fa);
      }
      continue;
    }
    for (size_t j = 0; j < st.blocks.size(); ++j) {
      const BlockP& bp = st.blocks[j];
      BlockSave& bs = sv.blocks[j];
      if (!save) A.off = mark0;
      bs.x = cur;
      bs.ln = A.alloc((size_t)N * C * es);
      bs.mean = (float*)A.alloc((size_t)N * sizeof(float));
      bs.rstd = (float*)A.alloc((size_t)N * sizeof(float));
      bs.u = A.alloc((size_t)N * 2 * C * es);
      // h = gelu(u): an operand of the pw2 weight gradient - re-formed from u by the one-launch small-C backward, read from
      // memory by the weight-gradient riders of the wider Blocks.  z: saved by nobody since round 4 (dgamma comes from the
      // raw pw2 weight gradient, kernels.h: cnn_gamma_finish): 7 -> 6 (4 for C <= 8) stored tensors of N C elements per Block
      const bool small_full = C <= 8 && c->fuse && !(c->nochain & 512);
      bs.h = small_full ? nullptr : A.alloc((size_t)N * 2 * C * es);
      bs.z = nullptr;
      TailLn tl{};
      if (tail_ln && j + 1 == st.blocks.size() && (C <= 8 || !(c->nochain & 1))) {
        if (si + 1 < cfg.n_stages) {          // the next stage's Downsample norm
          const StageP& nx = L.stages[si + 1];
          StageSave& nsv = c->ssave[si + 1];
          nsv.ln = A.alloc((size_t)N * C * es);
          nsv.mean = (float*)A.alloc((size_t)N * sizeof(float));
          nsv.rstd = (float*)A.alloc((size_t)N * sizeof(float));
          tl = TailLn{c->P(nx.ln_w), c->P(nx.ln_b), nsv.ln, nsv.mean, nsv.rstd, 0};
        } else if (C == L.H) {                // the final norm (model.py:759-762)
          c->fn_mean = (float*)A.alloc((size_t)N * sizeof(float));
          c->fn_rstd = (float*)A.alloc((size_t)N * sizeof(float));
          fn_y_tail = A.alloc((size_t)N * C * es);
          tl = TailLn{c->P(L.fn_w), c->P(L.fn_b), fn_y_tail, c->fn_mean, c->fn_rstd, 1};   // the transformer's stream: Res
        }
        tail_done = tl.w != nullptr;
      }
      if (C <= 8) {   // per-position kernel: the 1x1 convs are too narrow for an MFMA tile
        void* y = out_buf((size_t)N * C);
        RUN(block_small_fwd(s, dt, cur, c->P(bp.dw_w), c->P(bp.dw_b), c->P(bp.ln_w), c->P(bp.ln_b), c->WC(bp.c_pw1),
                            c->P(bp.pw1_b), c->WC(bp.c_pw2), c->P(bp.pw2_b), c->P(bp.gamma), save ? bs.ln : nullptr,
                            save ? bs.mean : nullptr, save ? bs.rstd : nullptr, save ? bs.u : nullptr, save ? bs.h : nullptr,
                            save ? bs.z : nullptr, y, B, Lcur, C, &tl));
        cur = y;
        continue;
      }
      if (c->fuse && !(c->nochain & 1)) {
        void* y = out_buf((size_t)N * C);
        BlockFwdArgs fa{cur, c->P(bp.dw_w), c->P(bp.dw_b), c->P(bp.ln_w), c->P(bp.ln_b), c->WC(bp.c_pw1p), c->P(bp.pw1_b),
                        c->WC(bp.c_pw2p), c->P(bp.pw2_b), c->P(bp.gamma), bs.ln, bs.mean, bs.rstd, bs.u, bs.h, bs.z, y,
                        B, Lcur, C, dbg_slot++, tl};
        if (!save) { fa.LN = nullptr; fa.mean = nullptr; fa.rstd = nullptr; fa.U = nullptr; fa.Hh = nullptr; fa.Z = nullptr; }
        RUN(block_fwd_fused(s, dt, fa));
        cur = y;
        continue;
      }
      RUN(dwln_fwd(s, dt, cur, c->P(bp.dw_w), c->P(bp.dw_b), c->P(bp.ln_w), c->P(bp.ln_b), bs.ln, bs.mean, bs.rstd, B,
                   Lcur, C));
      GemmArgs g1 = gargs(bs.ln, C, c->WC(bp.c_pw1), C, N, 2 * C, C, c->P(bp.pw1_b), bs.u, 2 * C);
      g1.Y2 = bs.h; g1.ldy2 = 2 * C;
      RUN(gemm_nt(s, dt, EPI_GELU, g1));
      void* y = out_buf((size_t)N * C);
      GemmArgs g2 = gargs(bs.h, 2 * C, c->WC(bp.c_pw2), 2 * C, N, C, 2 * C, c->P(bp.pw2_b), y, C);
      g2.Y2 = bs.z; g2.ldy2 = C; g2.R = cur; g2.ldr = C; g2.gamma = c->P(bp.gamma);
      RUN(gemm_nt(s, dt, EPI_SCALE_RESID, g2));
      cur = y;
    }
  }
  // final LayerNorm; channels-last means the reference's transpose (model.py:762) is free
  const int S = Lcur, N = B * S, H = L.H;
  c->fn_x = cur;
  if (tail_done) {   // evaluated by the last Block's launch; its buffers sit above that Block's (dead in inference) saves
    cur = fn_y_tail;
    tail_done = false;
  } else {
    if (!save) A.off = mark0;
    c->fn_mean = (float*)A.alloc((size_t)N * sizeof(float));
    c->fn_rstd = (float*)A.alloc((size_t)N * sizeof(float));
    void* y = out_buf((size_t)N * H);
    RUN(layernorm_fwd(s, dt, cur, c->P(L.fn_w), c->P(L.fn_b), y, c->fn_mean, c->fn_rstd, N, H, true));   // output: Res
    cur = y;
  }
  c->fn_y = cur;
  const size_t mark1 = A.off;
  bool qkv_done = false;   // the current layer's first half was computed by the previous layer's boundary launch
  int qkv_prerot = 0;      // ... and left q / k rotated (QkvFwdArgs::rope_cos)
  // XCD-team form: the whole stack in ONE launch (kernels.h: TLayerArgs; tlayer.hip).  Training sizes, the configuration the
  // chain kernels serve, bf16 attention; the saved tensors are the launch-per-operator path's, so the backward pass is unchanged.
  bool stack_done = false;
  if (save && c->tlayer && c->team != nullptr && !c->team_dead && c->fuse && c->fuse_ffn && !fp8 && N < 8192 &&
      !(c->nochain & (4 | 8)) && H == 256 && cfg.ffn_inter == 512 && cfg.n_heads == 4 && cfg.head_dim == 64 &&
      cfg.kv_compress == 64 && L.layers.size() >= 1 && L.layers.size() <= (size_t)TLAYER_MAX && (dt == DT_F32 || S > 256)) {
    TLayerArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.X0 = cur; ta.st = c->team; ta.stamps = c->tl_stamps; ta.rope_cos = c->rope_cos; ta.rope_sin = c->rope_sin;
    ta.seed_dev = c->seed_word; ta.seed = seed; ta.B = B; ta.S = S; ta.n_layers = (int)L.layers.size();
    ta.prerot = c->prerot ? 1 : 0;
    {
      const int stride = 8, W = 16;             // attn.hip: local_geom
      const int req = stride - (S - W) % stride;
      const int pl = req != stride ? req / 2 : 0, pr = req != stride ? req / 2 + (req % 2) : 0;
      ta.padL = pl; ta.nW = (S + pl + pr - W) / stride + 1;
    }
    ta.drop_thresh24 = dp.thr; ta.inv_keep = dp.inv_keep;
    const void* x = cur;
    for (size_t li = 0; li < L.layers.size(); ++li) {
      const LayerP& lp = L.layers[li];
      LayerSave& ls = c->lsave[li];
      const bool local = (li % 2 == 0);
      ls.x = const_cast<void*>(x);
      ls.xn = A.alloc((size_t)N * H * es);
      ls.m1 = (float*)A.alloc((size_t)N * sizeof(float));
      ls.r1 = (float*)A.alloc((size_t)N * sizeof(float));
      ls.qc = A.alloc((size_t)N * L.QC * es);
      ls.kv = A.alloc((size_t)N * L.KVW * es);
      ls.ao = A.alloc((size_t)N * H * es);
      ls.lse = local ? nullptr : (float*)A.alloc((size_t)B * cfg.n_heads * S * sizeof(float));
      {
        const size_t nt = (size_t)(S + 63) / 64;
        ls.keepbits = (!local && dp.thr && !c->attn_nobits)
                          ? (unsigned long long*)A.alloc((size_t)B * cfg.n_heads * nt * nt * 64 * sizeof(unsigned long long)) : nullptr;
      }
      ls.hn = A.alloc((size_t)N * H * es);
      ls.m2 = (float*)A.alloc((size_t)N * sizeof(float));
      ls.r2 = (float*)A.alloc((size_t)N * sizeof(float));
      ls.u = A.alloc((size_t)N * L.I2 * es);
      ls.g = A.alloc((size_t)N * (L.I2 / 2) * es);
      ls.h = A.alloc((size_t)N * H * es);
      void* y = A.alloc((size_t)N * H * es);
      ls.prerot = (!local && c->prerot) ? 1 : 0;
      TLayerItem& it = ta.it[li];
      it.Wqc = c->WC(lp.c_wqcp); it.Wkv = c->WC(lp.c_wkvp); it.Wo = c->WC(lp.c_wop); it.W1 = c->WC(lp.c_w1p); it.W2 = c->WC(lp.c_w2p);
      it.ln1w = c->P(lp.ln1_w); it.ln1b = c->P(lp.ln1_b); it.ln2w = c->P(lp.ln2_w); it.ln2b = c->P(lp.ln2_b);
      it.b1 = c->P(lp.b1); it.b2 = c->P(lp.b2);
      it.xn = ls.xn; it.qc = ls.qc; it.kv = ls.kv; it.ao = ls.ao; it.h = ls.h; it.hn = ls.hn; it.u = ls.u; it.g = ls.g; it.y = y;
      it.m1 = ls.m1; it.r1 = ls.r1; it.m2 = ls.m2; it.r2 = ls.r2; it.lse = ls.lse; it.keepbits = ls.keepbits;
      x = y;
    }
    if (dry || A.overflow) {
      stack_done = true;      // planning: the allocations above are the plan
    } else if (tlayer_fwd(s, dt, ta)) {
      stack_done = true;
    } else {
      c->err = "tlayer_fwd: not served (AMT_TLAYER=0 selects the launch-per-operator path)";
      return 1;
    }
    cur = const_cast<void*>(x);
  }
  for (size_t li = 0; !stack_done && li < L.layers.size(); ++li) {
    const LayerP& lp = L.layers[li];
    LayerSave& ls = c->lsave[li];
    const bool local = (li % 2 == 0);   // model.py:609-611: local layer first, then global
    if (!save) A.off = mark1;
    ls.x = cur;
    if (!qkv_done) {   // (otherwise allocated - and computed - by the previous layer's boundary launch)
      ls.xn = A.alloc((size_t)N * H * es);
      ls.m1 = (float*)A.alloc((size_t)N * sizeof(float));
      ls.r1 = (float*)A.alloc((size_t)N * sizeof(float));
      ls.qc = A.alloc((size_t)N * L.QC * es);
      ls.kv = A.alloc((size_t)N * L.KVW * es);
    }
    ls.ao = A.alloc((size_t)N * H * es);
    ls.lse = local ? nullptr : (float*)A.alloc((size_t)B * cfg.n_heads * S * sizeof(float));
    // keep bits of the global layer's attention-weight dropout, written by the forward, read by both backward roles
    // (kernels.h: AttnArgs::keepbits); the fp8 forward does not write them (its backward falls back to the hash)
    {
      const size_t nt = (size_t)(S + 63) / 64;
      // (allocated whether or not the fp8 forward is switched on: the switch may change between calls of one cached plan)
      ls.keepbits = (!local && save && dp.thr && !c->attn_nobits)
                        ? (unsigned long long*)A.alloc((size_t)B * cfg.n_heads * nt * nt * 64 * sizeof(unsigned long long)) : nullptr;
      if (fp8) ls.keepbits = nullptr;
    }
    ls.hn = A.alloc((size_t)N * H * es);
    ls.m2 = (float*)A.alloc((size_t)N * sizeof(float));
    ls.r2 = (float*)A.alloc((size_t)N * sizeof(float));
    ls.u = A.alloc((size_t)N * L.I2 * es);
    ls.g = A.alloc((size_t)N * (L.I2 / 2) * es);
    const bool chain_ok = c->fuse && H == 256 && cfg.ffn_inter == 512 && cfg.n_heads * cfg.head_dim == 256 &&
                          cfg.kv_compress == 64;   // == the predicate the weight copies are derived under (amt_create)
    auto qkv_args = [&](size_t l, const void* x) {
      const LayerP& q = L.layers[l];
      const LayerSave& t = c->lsave[l];
      QkvFwdArgs qa;
      memset(&qa, 0, sizeof(qa));
      qa.X = x; qa.ln_w = c->P(q.ln1_w); qa.ln_b = c->P(q.ln1_b); qa.Wqc = c->WC(q.c_wqcp); qa.Wkv = c->WC(q.c_wkvp);
      qa.XN = save ? t.xn : nullptr;        // LN1(x) is an operand of the backward pass only
      qa.mean = t.m1; qa.rstd = t.r1; qa.QC = t.qc; qa.KV = t.kv; qa.N = N; qa.dbg_slot = 96 + (int)l;
      // global layers: q and k leave the chain rotated (kernels.h: QkvFwdArgs::rope_cos), the attention kernels copy them.
      // Not with the fp8 forward (it stages from the raw projections); AMT_PREROT=0 keeps the raw form everywhere.
      if (l % 2 == 1 && !fp8 && c->prerot) { qa.rope_cos = c->rope_cos; qa.rope_sin = c->rope_sin; qa.rope_S = S; }
      return qa;
    };
    if (qkv_done) {
      qkv_done = false;                 // xn, qc, kv of this layer came out of the previous layer's boundary launch
      ls.prerot = qkv_prerot;
    } else if (chain_ok && !(c->nochain & 4)) {
      const QkvFwdArgs qa = qkv_args(li, cur);
      ls.prerot = qa.rope_cos != nullptr;
      RUN(qkv_fwd_fused(s, dt, qa));
    } else {
      ls.prerot = 0;
      RUN(layernorm_fwd(s, dt, cur, c->P(lp.ln1_w), c->P(lp.ln1_b), ls.xn, ls.m1, ls.r1, N, H));
      GemmArgs gq = gargs(ls.xn, H, c->WC(lp.c_wqc), H, N, L.QC, H, nullptr, ls.qc, L.QC);
      RUN(gemm_nt(s, dt, EPI_STORE, gq));
      const void* cptr = static_cast<const char*>(ls.qc) + (size_t)(L.QC - cfg.kv_compress) * es;
      GemmArgs gk = gargs(cptr, L.QC, c->WC(lp.c_wkv), cfg.kv_compress, N, L.KVW, cfg.kv_compress, nullptr, ls.kv, L.KVW);
      RUN(gemm_nt(s, dt, EPI_STORE, gk));
    }
    AttnArgs aa;
    memset(&aa, 0, sizeof(aa));
    aa.QC = ls.qc; aa.ldqc = L.QC; aa.KV = ls.kv; aa.ldkv = L.KVW; aa.O = ls.ao; aa.ldo = H; aa.LSE = ls.lse;
    aa.cos = c->rope_cos; aa.sin = c->rope_sin; aa.B = B; aa.S = S; aa.H = cfg.n_heads;
    aa.seed = seed; aa.seed_dev = c->seed_word; aa.stream = stream_attn((int)li); aa.drop_thresh24 = dp.thr; aa.inv_keep = dp.inv_keep;
    aa.keepbits = ls.keepbits;
    aa.prerot = local ? 0 : ls.prerot;
    if (local) RUN(attn_local_fwd(s, dt, aa));
    else if (fp8) RUN(attn_global_fwd_fp8(s, aa, c->amax + 8 * li, c->amax + 8 * li + 4));
    else RUN(attn_global_fwd(s, dt, aa));
    // h = x + attn . Wo^T ; out = h + FFN(LN2(h))
    ls.h = save ? A.alloc((size_t)N * H * es) : out_buf(0);
    void* y = save ? A.alloc((size_t)N * H * es) : out_buf(0);
    if (chain_ok && !(c->nochain & 8)) {
      FfnFwdArgs fa;
      memset(&fa, 0, sizeof(fa));
      fa.AO = ls.ao; fa.X = cur; fa.Wo = c->WC(lp.c_wop); fa.ln_w = c->P(lp.ln2_w); fa.ln_b = c->P(lp.ln2_b);
      fa.Hout = ls.h; fa.HN = ls.hn; fa.mean = ls.m2; fa.rstd = ls.r2; fa.N = N;
      fa.dbg_slot = 40 + (int)li;
      if (c->fuse_ffn) {
        fa.W1 = c->WC(lp.c_w1p); fa.b1 = c->P(lp.b1); fa.W2 = c->WC(lp.c_w2p); fa.b2 = c->P(lp.b2);
        fa.U = ls.u; fa.G = ls.g; fa.Y = y;
        fa.seed = seed; fa.seed_dev = c->seed_word; fa.stream = stream_ffn((int)li); fa.drop_thresh24 = dp.thr; fa.inv_keep = dp.inv_keep;
        // layer boundary: this layer's second half and the next layer's LN1 + q/kv projections in one launch
        // (training sizes; the inference path recycles its buffers per layer and keeps the two launches)
        if (save && li + 1 < L.layers.size() && N < 8192 && !(c->nochain & 4) && !(c->nochain & 4096)) {
          LayerSave& nx = c->lsave[li + 1];
          nx.xn = A.alloc((size_t)N * H * es);
          nx.m1 = (float*)A.alloc((size_t)N * sizeof(float));
          nx.r1 = (float*)A.alloc((size_t)N * sizeof(float));
          nx.qc = A.alloc((size_t)N * L.QC * es);
          nx.kv = A.alloc((size_t)N * L.KVW * es);
          const QkvFwdArgs qa = qkv_args(li + 1, y);
          qkv_prerot = qa.rope_cos != nullptr;
          if (!dry && !A.overflow) {
            if (!ffn_qkv_fwd_fused(s, dt, fa, qa)) { ffn_fwd_fused(s, dt, fa); qkv_fwd_fused(s, dt, qa); }
          }
          qkv_done = true;
        } else {
          RUN(if (!ffn_fwd_big(s, dt, fa, save)) ffn_fwd_fused(s, dt, fa));
        }
        cur = y;
        continue;
      }
      RUN(ffn_fwd_fused(s, dt, fa));   // h = x + ao . Wo^T, hn = LN2(h)
    } else {
      GemmArgs go = gargs(ls.ao, H, c->WC(lp.c_wo), H, N, H, H, nullptr, ls.h, H);
      go.R = cur; go.ldr = H;
      RUN(gemm_nt(s, dt, EPI_RESID, go));
      RUN(layernorm_fwd(s, dt, ls.h, c->P(lp.ln2_w), c->P(lp.ln2_b), ls.hn, ls.m2, ls.r2, N, H));
    }
    GemmArgs g1 = gargs(ls.hn, H, c->WC(lp.c_w1), H, N, L.I2, H, c->P(lp.b1), ls.g, L.I2 / 2);
    g1.Y2 = ls.u; g1.ldy2 = L.I2;
    RUN(gemm_nt(s, dt, EPI_GLU, g1));
    GemmArgs g2 = gargs(ls.g, L.I2 / 2, c->WC(lp.c_w2), L.I2 / 2, N, H, L.I2 / 2, c->P(lp.b2), y, H);
    g2.R = ls.h; g2.ldr = H;
    g2.seed = seed; g2.seed_dev = c->seed_word; g2.stream = stream_ffn((int)li); g2.drop_thresh24 = dp.thr; g2.inv_keep = dp.inv_keep;
    RUN(gemm_nt(s, dt, EPI_RESID, g2));
    cur = y;
  }
  // decoder (model.py:185-198)
  if (!save) A.off = mark1;
  c->hd_x = cur;
  c->hd_n = A.alloc((size_t)N * H * es);
  c->hd_mean = (float*)A.alloc((size_t)N * sizeof(float));
  c->hd_rstd = (float*)A.alloc((size_t)N * sizeof(float));
  c->logits_pad = (float*)A.alloc((size_t)N * L.VP * sizeof(float));
  RUN(layernorm_fwd(s, dt, cur, c->P(L.hd_lnw), c->P(L.hd_lnb), c->hd_n, c->hd_mean, c->hd_rstd, N, H));
  GemmArgs gh = gargs(c->hd_n, H, c->WC(L.c_hw), H, N, L.VP, H, c->head_bias, c->logits_pad, L.VP);
  RUN(gemm_nt(s, dt, EPI_STORE_F32, gh));
#undef RUN
  return 0;
}

__global__ void set_int_kernel(int* p, int v) { *p = v; }
__global__ void advance_seed_kernel(uint64_t* p, uint64_t delta) { *p += delta; }

// ---- "first half complete" published from INSIDE the launch sequence (run_backward phase 3; include/amt.h: amt_first_half_wait)
// The overlapped data-parallel step wants its first collective to start when the gradients of [fn_w, n) are complete, in
// the MIDDLE of the backward pass.  What the obvious forms cost on this platform, all measured at cfg2 with NO collective
// running (scratch/overlap_cost.py, scratch/feeder_cost.py, profiles/r04_notes.md; the one-call step: 3.01 ms):
//   * the pass cut into two host calls / two graphs with the communication stream waiting for an event of the compute
//     stream in between (phases 1 / 2):                                                          3.14-3.16 ms.  Of the
//     +0.13 ms, ~0.12 are that ONE cross-stream wait: any stream waiting for an event recorded on the busy compute stream
//     costs that much (an event record alone: +5 us; the compute stream waiting for ANOTHER stream's event: +20 us);
//   * one graph + a one-wave kernel on the communication stream that polls a device counter:    3.16 ms (a kernel resident
//     on a second queue slows the ~200 dependent launches of the first);
//   * one graph + hipStreamWaitValue32 on HSA signal memory (the command processor polls):      4.4 ms;
//   * external event-record nodes inside a capture: "disallowed in rocm" (torch 2.10); a captured RCCL call aborts its watchdog.
// So: the one-call launch sequence stays ONE graph (+8 us: the first range's record sums and this kernel), one of its nodes
// publishes a counter into MAPPED HOST memory, and the HOST polls that word before it issues the collective on the
// communication stream.  No stream ever waits for the compute stream.
__global__ void half_signal_kernel(unsigned* counter, unsigned* host_word, const float* loss_acc, float* tail_loss, float loss_mul) {
  if (tail_loss != nullptr) tail_loss[0] = loss_acc[0] * loss_mul;      // the rank's loss travels with the first range
  const unsigned v = counter[0] + 1u;                                    // (one thread, launches of a context are serialised)
  counter[0] = v;
  __threadfence_system();                                                // the gradients of earlier launches and the loss word: visible first
  __hip_atomic_store(host_word, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------ backward
// phase 3: the whole backward pass (the launches of phase 0) plus, right behind the first CNN Block chain - which carries the
// last riders of the range [L.fn_w, n_params) - the record sums of that range and a one-thread kernel that publishes "first
// half complete" on the device (half_signal_kernel); fused protocol only (grads_out == nullptr).
// phase 0: the whole backward pass.  phase 1 / 2: its two halves for the overlapped data-parallel step —
// 1 = loss, decoder, transformer and final CNN LayerNorm (their gradients = the flat range
// [L.fn_w, n_params), complete and unscaled when the call returns), 2 = the CNN stages (range [0, L.fn_w)).
// Both halves run the same host bookkeeping (buffer rotation, scratch regions, record offsets, riders)
// and only launch inside their own half, so 1 followed by 2 issues exactly the launches of 0.
// Gradient accumulation over minibatches (train.py:275-310: lax.scan, fp32 accumulation, / (grad_scale * steps)): call
// acc_index = 0 .. acc_count - 1.  Call 0 zeroes the context's accumulator, the loss accumulators and sets the flag; every
// call ADDS its gradients there (the weight-gradient kernels add with fp32 atomics, the record sums add as well); the last
// call's final pass writes accumulator / (scale * acc_count) to the caller's buffer and checks it.  No extra pass, no
// second gradient buffer.
int run_backward(amt_ctx* c, hipStream_t s, const void* audio, const void* labels, int B, int T, float scale,
                 bool train, uint64_t seed, float* grads_out, int* grads_valid, int phase = 0, int acc_index = 0,
                 int acc_count = 1) {
  const bool acc_first = acc_index == 0, acc_last = acc_index == acc_count - 1;
  const float out_mul = 1.f / (scale * (float)acc_count);
  const amt_config& cfg = c->cfg;
  const Layout& L = c->L;
  // Gradients are accumulated (fp32 atomics, record sums) in a buffer of the context whose base is shifted so
  // that the first transformer layer - and with it every leaf from CNN stage 2 on, all multiples of 16 floats -
  // starts on a 256-byte boundary, whatever the caller's pointer is: with the 64-float rows of the weight-gradient
  // tiles misaligned by 48 bytes (the flat layout puts the transformer at float offset 2 370 684 + 512) the
  // rider launches of the transformer chains took 4.5 us longer each (0.17 ms per step).  The last pass over the
  // gradients (1 / scale, finite check) writes them to the caller's buffer.
  float* grads = c->gacc + c->gacc_pad;
  Arena& A = c->arena;
  const DType dt = c->dt;
  const size_t es = c->esz;
  const bool dry = A.dry;
  bool live = (phase != 2);            // launches enabled (first half active unless phase == 2)
#define RUN(...) do { if (!dry && live && !A.overflow) { __VA_ARGS__; } } while (0)
  const Drop dp = drop_params(c, train);
  const int S = T / 320, N = B * S, H = L.H;
  PartialSink sink;
  sink.ws = c->partials;
  sink.grads_base = grads;
  sink.dry = dry || !live;             // launchers that take the sink account but do not launch when it is "dry"
  struct StopGuard { Arena& a; ~StopGuard() { a.stop = nullptr; } } stop_guard{A};
  A.stop = &sink.dry;
  if (dry) { c->part_descs.clear(); sink.descs = &c->part_descs; }

  int dbg_bwd = 0;   // micro-benchmark builds only (AMT_CHAIN_TIMING)
  auto G = [&](size_t off) { return grads + off; };
  std::vector<WgradArgs> pending, pending_skinny;
  // CNN weight gradients "late" (round-2 experiment, opt-in AMT_CNN_WGRAD_LATE=1): riders make every Block-backward
  // launch end with its rider workgroups (26.9 us against 14.2 us for the chain, profiles/r01_notes.md).  Here every CNN
  // weight gradient is queued instead, its operands (du, dy) stay in buffers of their own (~220 MB at cfg2) and the
  // batch runs after the pass as 9 grouped launches.  Measured: Block backward 29.4 -> 22.1 us per launch, but the
  // 64 x 64-tile weight-gradient kernel reaches only 87-100 TFLOP/s on the batch (9 x 36-41 us): 3.85-3.91 ms/step
  // against 3.77 with riders (profiles/r02_notes.md).  Riders stay the default at cfg2; at large N, where nothing rides any
  // more, the late batch is the default (late_for below).
  std::vector<WgradArgs> cnn_late;
  bool late_mode = false;
  auto wgrad = [&](const void* Gm, int ldg, const void* Am, int lda, int n, int M, int Kp, size_t w_off, int ldw,
                   long b_off, int m_store, const float* row_scale = nullptr) {
    WgradArgs w;
    memset(&w, 0, sizeof(w));
    w.row_scale = row_scale;
    w.G = Gm; w.ldg = ldg; w.A = Am; w.lda = lda; w.N = n; w.M = M; w.Kp = Kp;
    w.dW = G(w_off); w.ldw = ldw; w.dbias = b_off >= 0 ? G((size_t)b_off) : nullptr; w.M_store = m_store;
    if (late_mode) { cnn_late.push_back(w); return; }             // CNN stages: one batch of launches after the pass
    const int tiles = ((M + 63) / 64) * ((Kp + 63) / 64);
    if (tiles == 1 && n >= 16384) pending_skinny.push_back(w);   // per-split records (gemm.hip)
    else pending.push_back(w);                                    // grouped launch
  };
  // ---- scopes.  The backward pass is a chain of scopes (decoder, each transformer layer, final
  // LayerNorm, each CNN block, each downsample).  The data-gradient chain of a scope runs on the
  // main stream; its weight-gradient launches are queued and issued on the side stream when the
  // scope ends, so they overlap with the NEXT scopes' chain.  Scratch tensors alternate between two
  // regions and the residual-gradient buffers rotate over three, and scope j waits for the side
  // work of scope j-2 before it reuses that scope's region / buffer.
  const bool side_on = c->use_side && c->side != nullptr;
  hipStream_t ws = side_on ? c->side : s;
  // Riders are a small-N device: they fill the compute units a chain's ~250 workgroups leave idle.  From ~8 000 tokens on
  // the chains fill the chip themselves, a rider only delays its carrier, and the weight gradients run better as grouped
  // launches of their own: at 32 clips per GPU 8.59 -> 8.38 ms from taking the riders out, 8.25 with 512 workgroups per
  // grouped launch (profiles/r04_notes.md).  AMT_RIDER_MAX_TOKENS moves the switch (0: riders at every size).
  static const long rider_max_tokens = getenv("AMT_RIDER_MAX_TOKENS") ? atol(getenv("AMT_RIDER_MAX_TOKENS")) : 8192;
  const bool no_riders = rider_max_tokens > 0 && (long)B * (T / 320) >= rider_max_tokens;
  std::vector<hipEvent_t> side_done;
  size_t mark = 0;
  size_t scratch_used = 0;
  auto next_event = [&]() -> hipEvent_t {
    if (dry) { c->ev_needed++; return nullptr; }
    return c->ev_pool[c->ev_used++];
  };
  // Deferred weight gradients ("riders"): without the side stream, the weight-gradient problems of a
  // scope are not launched when the scope ends.  If the NEXT scope starts with a chain kernel that can
  // carry them (block_bwd_fused), they run in extra workgroups of that launch, off the dependent
  // chain and without a launch of their own; otherwise they are launched when the next scope begins.
  // The buffers they read stay valid for exactly one more scope (two alternating scratch regions,
  // four rotating residual-gradient buffers; "pass" scopes below stretch it by one short scope).
  std::vector<WgradArgs> deferred;
  // (without riders the grouped launches have the chip to themselves: 512 workgroups - two per compute unit - instead of the
  // 336 that suit a launch shared with a chain; at 32 clips per GPU 336 / 512 / 768: 8.38 / 8.25 / 8.56 ms, profiles/r04_notes.md)
  static const int flush_target_big = getenv("AMT_WGRAD_TARGET_LARGE") ? atoi(getenv("AMT_WGRAD_TARGET_LARGE")) : 512;
  // AMT_FLUSH_SCOPES=2 (without riders only): the deferred problems of TWO scopes go into one grouped launch - what scope
  // t - 2 queued reads the scratch region scope t is about to reuse, and is launched at the very beginning of scope t, before
  // anything of scope t is allocated or written; the residual-gradient buffers rotate over four.  Measured at 32 clips per GPU:
  // 14 launches of 106 us against 21 of 71 us, 7.878 against 7.863 ms - a transformer layer's five problems (140 tiles) are
  // already past the size where a grouped launch gets better with more tiles.  Default 1.
  static const int flush_scopes_env = getenv("AMT_FLUSH_SCOPES") ? atoi(getenv("AMT_FLUSH_SCOPES")) : 1;
  const int flush_scopes = no_riders ? std::max(1, std::min(2, flush_scopes_env)) : 1;
  int deferred_scopes = 0;
  auto flush_deferred = [&]() {
    const size_t per = no_riders ? 16 : 8;
    for (size_t i = 0; i < deferred.size(); i += per) {
      int cnt = (int)std::min<size_t>(per, deferred.size() - i);
      if (no_riders && cnt > 1) gemm_tn_wgrad_group(s, dt, deferred.data() + i, cnt, &sink, flush_target_big);
      else gemm_tn_wgrad_group(s, dt, deferred.data() + i, cnt, &sink);
    }
    deferred.clear();
    deferred_scopes = 0;
  };
  auto begin_scope = [&](bool carrier = false) {
    const size_t j = side_done.size();
    if (side_on && j >= 2 && side_done[j - 2] != nullptr) RUN((void)hipStreamWaitEvent(s, side_done[j - 2], 0));
    A.off = mark + (j & 1) * c->scratch_max;
    if (no_riders) {
      if (flush_scopes <= 1 || deferred_scopes >= flush_scopes) flush_deferred();
    } else if (!carrier || deferred.size() > 8) {
      flush_deferred();
    }
  };
  auto end_scope = [&]() {
    const size_t j = side_done.size();
    const size_t base = mark + (j & 1) * c->scratch_max;
    if (A.off > base && A.off - base > scratch_used) scratch_used = A.off - base;
    hipEvent_t done = nullptr;
    if (!side_on) {
      deferred.insert(deferred.end(), pending_skinny.begin(), pending_skinny.end());
      deferred.insert(deferred.end(), pending.begin(), pending.end());
      pending.clear();
      pending_skinny.clear();
      if (!deferred.empty()) deferred_scopes++;      // the age, in scopes, of the oldest problem that waits
    }
    if (!pending.empty() || !pending_skinny.empty()) {
      if (side_on) {
        hipEvent_t ev = next_event();
        RUN((void)hipEventRecord(ev, s));
        RUN((void)hipStreamWaitEvent(ws, ev, 0));
      }
      for (const WgradArgs& w : pending_skinny) gemm_tn_wgrad(ws, dt, w, &sink);
      for (size_t i = 0; i < pending.size(); i += 8) {
        int cnt = (int)std::min<size_t>(8, pending.size() - i);
        gemm_tn_wgrad_group(ws, dt, pending.data() + i, cnt, &sink);
      }
      pending.clear();
      pending_skinny.clear();
      if (side_on) {
        done = next_event();
        RUN((void)hipEventRecord(done, ws));
        if (dry) done = reinterpret_cast<hipEvent_t>(1);   // placeholder so the wait is counted as present
      }
    }
    side_done.push_back(done);
  };
  // "Pass" scopes (final CNN LayerNorm, Downsample backward: two small launches, no chain kernel that could carry
  // riders): they do not flush the deferred weight gradients, so those ride on the first Block-backward chain of
  // the NEXT stage together with the pass scope's own.  A pass scope works in the scratch region of the scope
  // that follows it (it does not advance the region parity), so the deferred problems' scratch operands stay
  // intact until that scope's launch has run them; the residual-gradient buffers rotate over FOUR for the same
  // reason.  The weight gradients a pass scope queues must not read its own scratch (they read the
  // residual-gradient buffer and saved forward tensors only).
  auto begin_pass_scope = [&]() {
    if (side_on) { begin_scope(); return; }
    A.off = mark + (side_done.size() & 1) * c->scratch_max;
  };
  auto end_pass_scope = [&]() {
    if (side_on) { end_scope(); return; }
    const size_t base = mark + (side_done.size() & 1) * c->scratch_max;
    if (A.off > base && A.off - base > scratch_used) scratch_used = A.off - base;
    deferred.insert(deferred.end(), pending_skinny.begin(), pending_skinny.end());
    deferred.insert(deferred.end(), pending.begin(), pending.end());
    pending.clear();
    pending_skinny.clear();
  };
  c->ev_used = 0;
  if (dry) c->ev_needed = 0;

  // Kernels, not hipMemsetAsync: under hipGraph replay (ROCm 7.2, MI355X) the captured memset nodes of
  // this step did not take effect from the second replay on (gradients accumulated over replays;
  // scratch/graph_dbg2.py), kernel nodes keep their stream order.  One launch for the three of them where it applies.
  bool began = !acc_first;   // later minibatches of an accumulated step add to what is there
  const bool fused_opt = grads_out == nullptr && !dry;     // the optimizer consumes (and zeroes) the accumulator itself
  if (acc_first && fused_opt) {                             // clean by the invariant: only the two loss words and the flag
    RUN(step_begin(s, grads, 0, c->loss_acc, grads_valid));
    began = true;
  } else if (acc_first) {
    RUN(began = step_begin(s, grads, L.n_params, c->loss_acc, grads_valid));
  }
  if (!began) {
    RUN(hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(1), 0, s, grads_valid, 1));
    RUN(fill_f32(s, grads, 0.f, L.n_params));
    RUN(fill_f32(s, c->loss_acc, 0.f, 2));
  }
  if (fused_opt) { c->fused_used = true; c->gacc_mul = out_mul; }

  // two ping-pong gradient buffers for the residual stream / CNN activations
  const size_t act_elems = (size_t)B * (T / 5) * 4;
  const size_t res_elems = act_elems > (size_t)N * H ? act_elems : (size_t)N * H;
  void* dbuf[4] = {A.alloc(res_elems * es), A.alloc(res_elems * es), A.alloc(res_elems * es), A.alloc(res_elems * es)};
  // dropout-masked copies of the transformer residual gradients rotate the same way: the copy written at
  // the end of scope t is read by scope t + 1 and by its weight-gradient riders inside scope t + 2
  void* dropbuf[4] = {nullptr, nullptr, nullptr, nullptr};
  if (dp.thr) for (int i = 0; i < 4; ++i) dropbuf[i] = A.alloc((size_t)N * H * es);
  int di = 0;
  mark = A.off;
  A.off = mark + 2 * c->scratch_max;     // both scratch regions are part of the plan
  if (A.off > A.peak) A.peak = A.off;
  // buffers that live until the end of the pass (operands of the late CNN weight gradients), behind the scratch regions
  size_t keep_off = mark + 2 * c->scratch_max;
  auto keep_alloc = [&](size_t bytes) -> void* {
    const size_t o = (keep_off + 255) & ~(size_t)255;
    keep_off = o + bytes;
    if (keep_off > A.peak) A.peak = keep_off;
    if (A.dry) return reinterpret_cast<void*>(o + 4096);
    if (keep_off > A.cap) { A.overflow = true; if (A.stop) *A.stop = true; return A.base; }
    return A.base + o;
  };
  begin_scope();

  // loss + dlogits
  void* dlog = A.alloc((size_t)N * L.VP * es);
  RUN(bce_loss(s, dt, c->logits_pad, L.VP, labels, c->loss_acc, dlog, L.VP, N, cfg.n_events, scale, B, c->labels_f16));
  // decoder backward
  wgrad(dlog, L.VP, c->hd_n, H, N, L.VP, H, L.hd_w, H, (long)L.hd_b, cfg.n_events);
  void* dhn = A.alloc((size_t)N * H * es);
  {
    GemmArgs g = gargs(dlog, L.VP, c->WC(L.c_hwt), L.VP, N, H, L.VP, nullptr, dhn, H);
    RUN(gemm_nt(s, dt, EPI_STORE, g));
  }
  void* dcur = dbuf[di];
  layernorm_bwd(s, dt, dhn, c->hd_x, c->hd_mean, c->hd_rstd, c->P(L.hd_lnw), nullptr, dcur, G(L.hd_lnw),
                    G(L.hd_lnb), N, H, &sink);
  end_scope();

  const void* pre_dropped = nullptr;
  for (int li = (int)L.layers.size() - 1; li >= 0; --li) {
    const LayerP& lp = L.layers[li];
    const LayerSave& ls = c->lsave[li];
    const bool local = (li % 2 == 0);
    const bool chain_ok = c->fuse && H == 256 && cfg.ffn_inter == 512 && cfg.n_heads * cfg.head_dim == 256 &&
                          cfg.kv_compress == 64;
    // the two backward chain launches of this layer carry the previous scope's weight gradients: the
    // smaller problems ride on lnout_bwd_fused, the larger ones on qkv_bwd_fused
    begin_scope(chain_ok && !side_on && !(c->nochain & 128) && !no_riders);
    std::vector<WgradArgs> riders_a, riders_b;
    if (!no_riders) {     // (without riders what is still deferred here waits for its grouped launch: begin_scope)
      for (const WgradArgs& w : deferred) {
        const int tiles = ((w.M + 63) / 64) * ((w.Kp + 63) / 64);
        static const int split_tiles = getenv("AMT_RIDER_SPLIT") ? atoi(getenv("AMT_RIDER_SPLIT")) : 24;
        (tiles < split_tiles ? riders_a : riders_b).push_back(w);
      }
      deferred.clear();
    }
    const void* dy = dcur;
    const void* dyd = dy;
    if (dp.thr) {
      if (pre_dropped != nullptr) {
        dyd = pre_dropped;            // written by the previous scope's qkv_bwd_fused
      } else {
        void* t = A.alloc((size_t)N * H * es);
        RUN(dropout_apply(s, dt, dy, t, (size_t)N * H, seed, stream_ffn(li), dp.thr, dp.inv_keep, c->seed_word));
        dyd = t;
      }
    }
    pre_dropped = nullptr;
    // FFN
    wgrad(dyd, H, ls.g, L.I2 / 2, N, H, L.I2 / 2, lp.w2, L.I2 / 2, (long)lp.b2, 0);
    void* du = A.alloc((size_t)N * L.I2 * es);
    const bool ffn_bwd_chain = chain_ok && c->fuse_ffn && !(c->nochain & 16) && !(c->nochain & 2048);
    void* dhn2 = nullptr;
    if (!ffn_bwd_chain) {
      {
        GemmArgs g = gargs(dyd, H, c->WC(lp.c_w2t), H, N, L.I2 / 2, H, nullptr, du, L.I2);
        g.U = ls.u; g.ldu = L.I2;
        RUN(gemm_nt(s, dt, EPI_GLU_BWD, g));
      }
      dhn2 = A.alloc((size_t)N * H * es);
      GemmArgs g = gargs(du, L.I2, c->WC(lp.c_w1t), L.I2, N, H, L.I2, nullptr, dhn2, H);
      RUN(gemm_nt(s, dt, EPI_STORE, g));
    }
    wgrad(du, L.I2, ls.hn, H, N, L.I2, H, lp.w1, H, (long)lp.b1, 0);
    void* dh = A.alloc((size_t)N * H * es);
    void* dao = A.alloc((size_t)N * H * es);
    if (ffn_bwd_chain) {
      // the whole second half of the layer in one launch (chain.hip: ffn_bwd_fused)
      FfnBwdArgs fb;
      memset(&fb, 0, sizeof(fb));
      fb.dYD = dyd; fb.dY = dy; fb.U = ls.u; fb.W2t = c->WC(lp.c_w2tp); fb.W1t = c->WC(lp.c_w1tp); fb.Wot = c->WC(lp.c_wotp);
      fb.Hres = ls.h; fb.mean = ls.m2; fb.rstd = ls.r2; fb.ln_w = c->P(lp.ln2_w); fb.dU = du; fb.dH = dh; fb.dAO = dao; fb.N = N;
      ffn_bwd_fused(s, dt, fb, G(lp.ln2_w), G(lp.ln2_b), &sink, riders_a.data(), (int)riders_a.size());
      wgrad(dh, H, ls.ao, H, N, H, H, lp.wo, H, -1, 0);
    } else if (chain_ok && !(c->nochain & 16)) {
      LnOutBwdArgs la;
      memset(&la, 0, sizeof(la));
      la.dHN = dhn2; la.Hres = ls.h; la.mean = ls.m2; la.rstd = ls.r2; la.ln_w = c->P(lp.ln2_w); la.dY = dy;
      la.Wot = c->WC(lp.c_wotp); la.dH = dh; la.dAO = dao; la.N = N; la.dbg_slot = 0;
      lnout_bwd_fused(s, dt, la, G(lp.ln2_w), G(lp.ln2_b), &sink, riders_a.data(), (int)riders_a.size());
      wgrad(dh, H, ls.ao, H, N, H, H, lp.wo, H, -1, 0);
    } else {
      if (!riders_a.empty()) gemm_tn_wgrad_group(s, dt, riders_a.data(), (int)riders_a.size(), &sink);
      layernorm_bwd(s, dt, dhn2, ls.h, ls.m2, ls.r2, c->P(lp.ln2_w), dy, dh, G(lp.ln2_w), G(lp.ln2_b), N, H, &sink);
      // attention output projection
      wgrad(dh, H, ls.ao, H, N, H, H, lp.wo, H, -1, 0);
      GemmArgs g = gargs(dh, H, c->WC(lp.c_wot), H, N, H, H, nullptr, dao, H);
      RUN(gemm_nt(s, dt, EPI_STORE, g));
    }
    void* dqc = A.alloc((size_t)N * L.QC * es);
    void* dkv = A.alloc((size_t)N * L.KVW * es);
    float* delta = (float*)A.alloc((size_t)B * cfg.n_heads * S * sizeof(float));
    AttnBwdArgs ab;
    memset(&ab, 0, sizeof(ab));
    ab.f.QC = ls.qc; ab.f.ldqc = L.QC; ab.f.KV = ls.kv; ab.f.ldkv = L.KVW; ab.f.O = ls.ao; ab.f.ldo = H;
    ab.f.LSE = ls.lse; ab.f.cos = c->rope_cos; ab.f.sin = c->rope_sin; ab.f.B = B; ab.f.S = S; ab.f.H = cfg.n_heads;
    ab.f.seed = seed; ab.f.seed_dev = c->seed_word; ab.f.stream = stream_attn(li); ab.f.drop_thresh24 = dp.thr; ab.f.inv_keep = dp.inv_keep;
    ab.f.keepbits = local ? nullptr : ls.keepbits;
    ab.f.prerot = local ? 0 : ls.prerot;
    ab.dO = dao; ab.lddo = H; ab.dQC = dqc; ab.lddqc = L.QC; ab.dKV = dkv; ab.lddkv = L.KVW; ab.delta = delta;
    if (local) RUN(attn_local_bwd(s, dt, ab)); else RUN(attn_global_bwd(s, dt, ab));
    const int ckv = cfg.kv_compress;
    const void* cptr = static_cast<const char*>(ls.qc) + (size_t)(L.QC - ckv) * es;
    di = (di + 1) % 4;
    void* dx = dbuf[di];
    if (chain_ok && !(c->nochain & 32)) {
      QkvBwdArgs qa;
      memset(&qa, 0, sizeof(qa));
      qa.dKV = dkv; qa.dQC = dqc; qa.Wkvt = c->WC(lp.c_wkvtp); qa.Wqct = c->WC(lp.c_wqctp);
      qa.X = ls.x; qa.mean = ls.m1; qa.rstd = ls.r1; qa.ln_w = c->P(lp.ln1_w); qa.dH = dh; qa.dX = dx; qa.N = N;
      if (dp.thr && li > 0 && !(c->nochain & 256)) {   // the next scope's FFN-output dropout mask is applied here (saves its dropout_apply launch)
        void* dxd = dropbuf[di];
        qa.dXD = dxd; qa.seed = seed; qa.seed_dev = c->seed_word; qa.stream = stream_ffn(li - 1); qa.drop_thresh24 = dp.thr; qa.inv_keep = dp.inv_keep;
        pre_dropped = dxd;
      }
      qkv_bwd_fused(s, dt, qa, G(lp.ln1_w), G(lp.ln1_b), &sink, riders_b.data(), (int)riders_b.size());
      wgrad(dkv, L.KVW, cptr, L.QC, N, L.KVW, ckv, lp.wkv, ckv, -1, 0);
      wgrad(dqc, L.QC, ls.xn, H, N, L.QC, H, lp.wqc, H, -1, 0);
    } else {
      if (!riders_b.empty()) gemm_tn_wgrad_group(s, dt, riders_b.data(), (int)riders_b.size(), &sink);
      // k/v up-projection
      wgrad(dkv, L.KVW, cptr, L.QC, N, L.KVW, ckv, lp.wkv, ckv, -1, 0);
      {
        void* dcp = static_cast<char*>(dqc) + (size_t)(L.QC - ckv) * es;
        GemmArgs g = gargs(dkv, L.KVW, c->WC(lp.c_wkvt), L.KVW, N, ckv, L.KVW, nullptr, dcp, L.QC);
        RUN(gemm_nt(s, dt, EPI_STORE, g));
      }
      // q / kv-down projection
      wgrad(dqc, L.QC, ls.xn, H, N, L.QC, H, lp.wqc, H, -1, 0);
      void* dxn = A.alloc((size_t)N * H * es);
      {
        GemmArgs g = gargs(dqc, L.QC, c->WC(lp.c_wqct), L.QC, N, H, L.QC, nullptr, dxn, H);
        RUN(gemm_nt(s, dt, EPI_STORE, g));
      }
      layernorm_bwd(s, dt, dxn, ls.x, ls.m1, ls.r1, c->P(lp.ln1_w), dh, dx, G(lp.ln1_w), G(lp.ln1_b), N, H, &sink);
    }
    dcur = dx;
    end_scope();
  }
  // final CNN LayerNorm
  {
    begin_pass_scope();
    di = (di + 1) % 4;
    void* dx = dbuf[di];
    layernorm_bwd(s, dt, dcur, c->fn_x, c->fn_mean, c->fn_rstd, c->P(L.fn_w), nullptr, dx, G(L.fn_w), G(L.fn_b), N,
                      H, &sink);
    dcur = dx;
    end_pass_scope();
  }
  // ---- boundary between the two halves: every gradient of [L.fn_w, n_params) is complete here
  if (no_riders) flush_deferred();      // (the last transformer scopes' problems: the first range ends here)
  if (dry) c->n_part_desc_half = (int)c->part_descs.size();
  if (phase == 1) {
    flush_deferred();   // the first transformer layer's weight gradients (in the one-call step they ride on stage 6)
    RUN(reduce_partials(s, c->part_desc_dev, c->n_part_desc_half, c->partials, grads));
    if (acc_last && grads_out != nullptr) RUN(grads_finalize(s, grads + L.fn_w, grads_out + L.fn_w, L.n_params - L.fn_w, out_mul, grads_valid));
    return 0;
  }
  if (phase == 2) { live = true; sink.dry = dry || A.overflow; deferred.clear(); }   // phase 1 ran the deferred problems itself (none of them takes sink records)
  bool signalled = phase != 3;
  auto signal_first_half = [&]() {     // phase 3: every launch that adds to [L.fn_w, n_params) has been issued
    // (called right behind the first CNN Block's launch and BEFORE its end_scope: the riders of the last transformer scopes
    // have travelled on that launch - or were flushed when the scope began - and the Block's own weight gradients, still
    // pending, belong to the second range)
    if (signalled) return;
    signalled = true;
    RUN(reduce_partials(s, c->part_desc_dev, c->n_part_desc_half, c->partials, grads));
    if (acc_last)
      RUN(hipLaunchKernelGGL(half_signal_kernel, dim3(1), dim3(1), 0, s, c->half_sig, c->half_host_dev, c->loss_acc, grads + L.n_params,
                             1.f / (float)acc_count));
  };
  // CNN stages, last to first
  // Default at large N (no riders, from 8192 tokens on): ALL CNN weight gradients late.  A Block's pair of 1x1-conv problems is
  // 16 tiles - as a launch of its own that is 32 token splits of 500 tokens, four stages and a 16-KB atomic tile per workgroup;
  // eight problems (four Blocks) per launch: 32 clips per GPU 8.065 -> 7.94 ms (49 -> 25 grouped launches; 512 / 1024 / 2048
  // workgroups per late launch: 7.94 / 7.99 / 7.94; only C <= 128: 7.96; profiles/r04_notes.md); sixteen problems and 768
  // workgroups: 7.88.  At cfg2 riders stay (AMT_CNN_WGRAD_LATE=1 / 32 / 64 there: 3.126 / 2.985 / 3.000 against 2.976 ms).
  const int late_max_c = c->cnn_late_env ? c->cnn_late_max_c : (no_riders ? (1 << 20) : 0);
  auto late_for = [&](int si) { return si >= 0 && !side_on && L.stages[si].C <= late_max_c; };
  int Lcur = S;
  for (int si = cfg.n_stages - 1; si >= 0; --si) {
    const StageP& st = L.stages[si];
    const StageSave& sv = c->ssave[si];
    const int C = st.C;
    const int Nn = B * Lcur;
    late_mode = late_for(si);
    for (int j = (int)st.blocks.size() - 1; j >= 0; --j) {
      const BlockP& bp = st.blocks[j];
      const BlockSave& bs = sv.blocks[j];
      const bool carrier = c->fuse && C >= 16 && !side_on && !(c->nochain & 2) && !(c->nochain & 64) && !no_riders;
      begin_scope(carrier);
      const void* dy = dcur;
      if (C <= 8 && c->fuse && !(c->nochain & 512)) {
        // stages 0 / 1: the whole Block backward, weight gradients included, in one launch
        di = (di + 1) % 4;
        void* dx = late_mode ? keep_alloc((size_t)Nn * C * es) : dbuf[di];
        SmallBwdArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.dY = dy; sa.X = bs.x; sa.U = bs.u; sa.LN = bs.ln; sa.mean = bs.mean; sa.rstd = bs.rstd;
        sa.w2tg = c->WC(bp.c_pw2tg); sa.w1t = c->WC(bp.c_pw1t); sa.wdw = c->P(bp.dw_w); sa.bdw = c->P(bp.dw_b);
        sa.lnw = c->P(bp.ln_w); sa.gamma = c->P(bp.gamma); sa.dX = dx;
        sa.dW1 = G(bp.pw1_w); sa.db1 = G(bp.pw1_b); sa.dW2 = G(bp.pw2_w); sa.db2 = G(bp.pw2_b); sa.dlnw = G(bp.ln_w);
        sa.dlnb = G(bp.ln_b); sa.dwdw = G(bp.dw_w); sa.dbdw = G(bp.dw_b);
        sa.B = B; sa.L = Lcur; sa.C = C;
        block_small_bwd_full(s, dt, sa, &sink);
        dcur = dx;
        signal_first_half();
        end_scope();
        continue;
      }
      // layer scale folded away: dz = gamma * dy is never materialised.  The data-gradient GEMM uses a gamma-scaled copy of
      // pw2^T; the pw2 weight / bias gradients are accumulated RAW (sum dy h, sum dy) and cnn_gamma_finish, at the end of
      // the pass, forms dgamma = <W2, dW2raw> + b2 db2raw from them and scales them by gamma (kernels.h): no z anywhere.
      wgrad(dy, C, bs.h, 2 * C, Nn, C, 2 * C, bp.pw2_w, 2 * C, (long)bp.pw2_b, 0);
      void* du = late_mode ? keep_alloc((size_t)Nn * 2 * C * es) : A.alloc((size_t)Nn * 2 * C * es);
      if (c->fuse && C >= 16 && !(c->nochain & 2)) {
        wgrad(du, 2 * C, bs.ln, C, Nn, 2 * C, C, bp.pw1_w, C, (long)bp.pw1_b, 0);
        di = (di + 1) % 4;
        void* dx = late_mode ? keep_alloc((size_t)Nn * C * es) : dbuf[di];   // late: the next Block's dy operand must survive
        BlockBwdArgs ba;
        memset(&ba, 0, sizeof(ba));
        ba.dY = dy; ba.X = bs.x; ba.U = bs.u; ba.Z = nullptr; ba.W2tg = c->WC(bp.c_pw2tgp); ba.W1t = c->WC(bp.c_pw1tp);
        ba.wdw = c->P(bp.dw_w); ba.bdw = c->P(bp.dw_b); ba.lnw = c->P(bp.ln_w); ba.mean = bs.mean; ba.rstd = bs.rstd;
        ba.dU = du; ba.dX = dx; ba.B = B; ba.L = Lcur; ba.C = C; ba.dbg_slot = 64 + dbg_bwd++;
        block_bwd_fused(s, dt, ba, G(bp.ln_w), G(bp.ln_b), G(bp.dw_w), G(bp.dw_b), nullptr, &sink,
                        deferred.data(), (int)deferred.size());
        deferred.clear();
        dcur = dx;
        signal_first_half();
        end_scope();
        continue;
      }
      void* dln = A.alloc((size_t)Nn * C * es);
      if (C <= 8) {
        RUN(block_small_bwd(s, dt, dy, bs.u, c->WC(bp.c_pw2tg), c->WC(bp.c_pw1t), du, dln, (long)Nn, C));
      } else {
        GemmArgs g = gargs(dy, C, c->WC(bp.c_pw2tg), C, Nn, 2 * C, C, nullptr, du, 2 * C);
        g.U = bs.u; g.ldu = 2 * C;
        RUN(gemm_nt(s, dt, EPI_GELU_BWD, g));
        GemmArgs g2 = gargs(du, 2 * C, c->WC(bp.c_pw1t), 2 * C, Nn, C, 2 * C, nullptr, dln, C);
        RUN(gemm_nt(s, dt, EPI_STORE, g2));
      }
      wgrad(du, 2 * C, bs.ln, C, Nn, 2 * C, C, bp.pw1_w, C, (long)bp.pw1_b, 0);
      di = (di + 1) % 4;
      void* dx = late_mode ? keep_alloc((size_t)Nn * C * es) : dbuf[di];
      dwblock_bwd(s, dt, dln, bs.x, dy, nullptr, c->P(bp.dw_w), c->P(bp.dw_b), bs.mean, bs.rstd, c->P(bp.ln_w), dx,
                  G(bp.ln_w), G(bp.ln_b), G(bp.dw_w), G(bp.dw_b), nullptr, B, Lcur, C, &sink);
      dcur = dx;
      signal_first_half();
      end_scope();
    }
    if (si > 0) {
      begin_pass_scope();
      const int Cin = st.Cin, Lin = Lcur * 2;
      wgrad(dcur, C, sv.ln, 2 * Cin, Nn, C, 2 * Cin, st.conv_w, 2 * Cin, (long)st.conv_b, 0);
      void* dln = A.alloc((size_t)Nn * 2 * Cin * es);
      {
        GemmArgs g = gargs(dcur, C, c->WC(st.c_wt), C, Nn, 2 * Cin, C, nullptr, dln, 2 * Cin);
        RUN(gemm_nt(s, dt, EPI_STORE, g));
      }
      di = (di + 1) % 4;
      void* dx = late_for(si - 1) ? keep_alloc((size_t)B * Lin * Cin * es) : dbuf[di];   // dy operand of the stage below
      layernorm_bwd(s, dt, dln, sv.x, sv.mean, sv.rstd, c->P(st.ln_w), nullptr, dx, G(st.ln_w), G(st.ln_b), B * Lin,
                        Cin, &sink);
      dcur = dx;
      Lcur = Lin;
      end_pass_scope();
    } else {
      stem_bwd(s, dt, audio, dcur, c->P(st.conv_w), c->P(st.conv_b), c->P(st.ln_w), G(st.conv_w), G(st.conv_b),
               G(st.ln_w), G(st.ln_b), B, T, &sink);
    }
  }
  flush_deferred();
  signal_first_half();     // (a model without Blocks)
  late_mode = false;
  {   // the CNN weight gradients, 8 problems per launch (16 at large N: 8 / 12 / 16 problems at 512 workgroups 7.943 / 7.906 /
      // 7.942 ms, 16 at 768 / 1024: 7.882 / 7.891), many token splits (the launches are large and independent)
    static const int late_target_env = getenv("AMT_LATE_WGRAD_WGS") ? atoi(getenv("AMT_LATE_WGRAD_WGS")) : 0;
    const int late_target = late_target_env > 0 ? late_target_env : (no_riders ? 768 : 1024);
    static const int late_group = getenv("AMT_LATE_WGRAD_GROUP") ? std::max(1, std::min(16, atoi(getenv("AMT_LATE_WGRAD_GROUP")))) : 16;
    const size_t per = no_riders ? (size_t)late_group : 8;
    for (size_t i = 0; i < cnn_late.size(); i += per) {
      const int cnt = (int)std::min<size_t>(per, cnn_late.size() - i);
      gemm_tn_wgrad_group(s, dt, cnn_late.data() + i, cnt, &sink, late_target);
    }
  }
  // train.py:319-322: grads / grad_scale, all-finite flag (the flag was set to 1 when the pass began)
  // join: the main stream waits for the side work of the last two scopes (older ones were waited for
  // when their regions were reused)
  if (side_on) {
    for (size_t j = side_done.size() >= 2 ? side_done.size() - 2 : 0; j < side_done.size(); ++j)
      if (side_done[j] != nullptr) RUN((void)hipStreamWaitEvent(s, side_done[j], 0));
  }
  if (dry && scratch_used > c->scratch_max) c->scratch_max = scratch_used;
  c->part_total = sink.off;
  if (phase == 2 || phase == 3) {
    RUN(reduce_partials(s, c->part_desc_dev + c->n_part_desc_half, c->n_part_desc - c->n_part_desc_half, c->partials, grads));
    if (acc_last) RUN(cnn_gamma_finish(s, dt, c->params, grads, c->gamma_desc_dev, c->n_gamma_desc, c->gamma_rows));
    if (acc_last && grads_out != nullptr) RUN(grads_finalize(s, grads, grads_out, L.fn_w, out_mul, grads_valid));
  } else {
    RUN(reduce_partials(s, c->part_desc_dev, c->n_part_desc, c->partials, grads));
    // layer-scale gradients from the raw pw2 gradients, which it then scales (kernels.h); once, over the accumulated sums
    if (acc_last) RUN(cnn_gamma_finish(s, dt, c->params, grads, c->gamma_desc_dev, c->n_gamma_desc, c->gamma_rows));
    if (acc_last && grads_out != nullptr) RUN(grads_finalize(s, grads, grads_out, L.n_params, out_mul, grads_valid));
  }
  // keep the fused protocol's invariant (accumulator clean between steps) when the two protocols are mixed on one context
  if (acc_last && grads_out != nullptr && c->fused_used) RUN(fill_f32(s, grads, 0.f, L.n_params));
#undef RUN
  return 0;
}

int plan(amt_ctx* c, int B, int T, int train_mode, hipStream_t s) {
  // train_mode: 0 = inference (recycled buffers), 1 = training (everything saved + backward scratch)
  c->arena.overflow = false;
  if (c->plan_B == B && c->plan_T == T && c->plan_train == train_mode && c->arena.base) return 0;
  for (const amt_ctx::PlanRec& r : c->plans) {
    if (r.B != B || r.T != T || r.train != train_mode) continue;
    if (c->arena.cap < r.peak || c->partials_cap < r.part_total) continue;   // (cannot happen while both only grow; re-plan rather than run in a workspace that is too small)
    // planned before: the arena and the record workspace are at least as large as this plan needs (they only
    // grow), its descriptors are still on the device: nothing is allocated, no captured graph is invalidated
    c->scratch_max = r.scratch_max;
    c->part_total = r.part_total;
    c->part_desc_dev = r.desc_dev;
    c->n_part_desc = r.n_desc;
    c->n_part_desc_half = r.n_half;
    c->ev_needed = r.ev_needed;
    c->arena.off = 0;
    c->arena.peak = 0;
    c->plan_B = B; c->plan_T = T; c->plan_train = train_mode;
    return 0;
  }
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &cs);
  if (cs != hipStreamCaptureStatusNone) {
    c->err = "workspace for this (B, T) is not planned yet: run once outside stream capture";
    return 1;
  }
  if (train_mode == 1 && c->use_side && c->side == nullptr)
    HIPCHK(c, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
  Arena dryA;
  dryA.dry = true;
  Arena keep = c->arena;
  c->arena = dryA;
  c->part_descs.clear();
  c->part_total = 0;
  c->n_part_desc_half = 0;
  c->ev_needed = 0;
  run_forward(c, s, nullptr, B, T, train_mode == 1, true, 0);
  if (train_mode == 1) {
    // pass 1 measures the per-scope scratch, pass 2 lays out the two alternating scratch regions
    c->scratch_max = 0;
    const size_t fwd_end = c->arena.off;
    run_backward(c, s, nullptr, nullptr, B, T, 1.f, true, 0, nullptr, nullptr);
    c->arena.off = fwd_end;
    c->arena.peak = fwd_end;
    run_backward(c, s, nullptr, nullptr, B, T, 1.f, true, 0, nullptr, nullptr);
  }
  const size_t need = c->arena.peak + 4096;
  c->arena = keep;
  amt_ctx::PlanRec rec{B, T, train_mode, need, c->scratch_max, c->part_total, c->ev_needed, nullptr, 0, 0};
  if (train_mode == 1) {
    while (c->ev_pool.size() < c->ev_needed + 8) {
      hipEvent_t e;
      HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
      c->ev_pool.push_back(e);
    }
    if (c->part_total > c->partials_cap) {
      HIPCHK(c, hipDeviceSynchronize());
      if (c->partials) (void)hipFree(c->partials);
      c->partials = nullptr;
      c->partials_cap = 0;
      c->plan_gen++;                    // graphs captured earlier hold the old record workspace
      HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->partials), c->part_total * sizeof(float)));
      c->partials_cap = c->part_total;
    }
    rec.n_desc = (int)c->part_descs.size();
    rec.n_half = c->n_part_desc_half;
    if (rec.n_desc > 0) {   // owned by the plan record until the context is destroyed
      HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&rec.desc_dev), c->part_descs.size() * sizeof(PartDesc)));
      HIPCHK(c, hipMemcpy(rec.desc_dev, c->part_descs.data(), c->part_descs.size() * sizeof(PartDesc), hipMemcpyHostToDevice));
    }
    c->part_desc_dev = rec.desc_dev;
    c->n_part_desc = rec.n_desc;
  }
  if (need > c->arena.cap) {
    HIPCHK(c, hipDeviceSynchronize());
    if (c->arena.base) (void)hipFree(c->arena.base);
    c->arena.base = nullptr;
    c->arena.cap = 0;
    c->plan_gen++;                      // graphs captured earlier hold pointers into the old arena
    const hipError_t ea = hipMalloc(reinterpret_cast<void**>(&c->arena.base), need);
    if (ea != hipSuccess) {             // the plan is NOT recorded: give its descriptors back and point at no plan
      if (rec.desc_dev) (void)hipFree(rec.desc_dev);
      c->part_desc_dev = nullptr;
      c->n_part_desc = c->n_part_desc_half = 0;
      c->plan_B = c->plan_T = 0; c->plan_train = -1;
      c->err = std::string("hipMalloc(workspace): ") + hipGetErrorString(ea);
      return 1;
    }
    c->arena.cap = need;
  }
  // the record is kept only now that every allocation it describes exists: after a failed hipMalloc above the next
  // call for this shape plans again and reports the out-of-memory error instead of running in a missing workspace
  for (auto it = c->plans.begin(); it != c->plans.end();) {
    if (it->B == B && it->T == T && it->train == train_mode) {
      if (it->desc_dev && it->desc_dev != rec.desc_dev) {     // an older plan of the same shape: its descriptors go with it
        (void)hipDeviceSynchronize();                          // (a graph captured for it may still be replaying them)
        (void)hipFree(it->desc_dev);
        c->plan_gen++;
      }
      it = c->plans.erase(it);
    } else {
      ++it;
    }
  }
  c->plans.push_back(rec);
  c->arena.off = 0;
  c->arena.peak = 0;
  c->arena.dry = false;
  c->arena.overflow = false;
  c->plan_B = B; c->plan_T = T; c->plan_train = train_mode;
  return 0;
}

int check_shape(amt_ctx* c, int B, int T) {
  if (B <= 0) { c->err = "B must be positive"; return 1; }
  if (T <= 0 || T % 320 != 0) { c->err = "T must be a positive multiple of 320 (5 * 2^6 samples per frame)"; return 1; }
  if (T / 320 < c->cfg.local_window) { c->err = "clip shorter than one local-attention window (16 frames)"; return 1; }
  return 0;
}

int refresh(amt_ctx* c, hipStream_t s) {
  const Layout& L = c->L;
  convert_batch(s, c->dt, c->params, c->wc, c->conv_dev, c->conv_prefix_dev, (int)L.conv.size(), c->conv_tiles);
  // padded head bias
  copy_pad_f32(s, c->params + L.hd_b, c->head_bias, c->cfg.n_events, L.VP);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { c->err = std::string("refresh_weights: ") + hipGetErrorString(e); return 1; }
  return 0;
}

}  // namespace

// ===================================================================================== C ABI
extern "C" {

void amt_default_config(amt_config* cfg) {
  memset(cfg, 0, sizeof(*cfg));
  cfg->n_stages = 7;
  const int depths[7] = {3, 3, 3, 3, 3, 21, 3};
  for (int i = 0; i < 7; ++i) { cfg->dims[i] = 4 << i; cfg->depths[i] = depths[i]; }
  cfg->n_layer_pairs = 8;
  cfg->n_heads = 4;
  cfg->head_dim = 64;
  cfg->kv_compress = 64;
  cfg->ffn_inter = 512;
  cfg->n_events = 90;
  cfg->local_window = 16;
  cfg->dropout_rate = 0.1f;
  cfg->act_dtype = AMT_BF16;
}

uint64_t amt_param_count(const amt_config* cfg) {
  std::string e;
  if (!validate(*cfg, e)) return 0;
  return build_layout(*cfg).n_params;
}

int32_t amt_layout_count(const amt_config* cfg) {
  std::string e;
  if (!validate(*cfg, e)) return -1;
  return (int32_t)build_layout(*cfg).segs.size();
}

int32_t amt_layout_segment(const amt_config* cfg, int32_t index, amt_segment* out) {
  std::string e;
  if (!validate(*cfg, e)) return 1;
  Layout L = build_layout(*cfg);
  if (index < 0 || index >= (int32_t)L.segs.size()) return 1;
  const Seg& sg = L.segs[index];
  memset(out, 0, sizeof(*out));
  snprintf(out->name, sizeof(out->name), "%s", sg.name.c_str());
  out->offset = sg.off;
  out->rows = sg.rows;
  out->cols = sg.cols;
  out->lr_depth = sg.lr_depth;
  return 0;
}

int32_t amt_num_frames(const amt_config* cfg, int32_t T) {
  (void)cfg;
  return (T > 0 && T % 320 == 0) ? T / 320 : -1;
}

const char* amt_last_error(const amt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int32_t amt_create(const amt_config* cfg, int32_t device, amt_ctx** out) {
  *out = nullptr;
  std::string e;
  if (!validate(*cfg, e)) { g_create_error = e; return 1; }
  hipError_t he = hipSetDevice(device);
  if (he != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(he); return 1; }
  amt_ctx* c = new amt_ctx();
  c->cfg = *cfg;
  c->L = build_layout(*cfg);
  c->dt = cfg->act_dtype == AMT_F32 ? DT_F32 : DT_BF16;
  c->esz = c->dt == DT_F32 ? 4 : 2;
  c->device = device;
  // measured on MI355X (profiles/r01_notes.md): running the weight-gradient launches on a side
  // stream does not help at B = 8 x 10 s (9.47 vs 9.09 ms/step): the chain kernels already occupy
  // every CU and slow down when sharing them.  Kept as an opt-in for larger per-GPU batches.
  c->use_side = getenv("AMT_SIDE_STREAM") != nullptr;
  c->fuse = !(getenv("AMT_FUSE") && atoi(getenv("AMT_FUSE")) == 0);
  c->attn_nobits = getenv("AMT_ATTN_BITS") && atoi(getenv("AMT_ATTN_BITS")) == 0;
  c->nochain = getenv("AMT_NOCHAIN") ? atoi(getenv("AMT_NOCHAIN")) : 0;
  // whole second half of a layer (output projection, LN2, W1 / GLU, W2, dropout, residual) in one launch.  Round 1
  // measured it slower and left it off; with 16-token forward tiles and the boundary cost understood (it is the write-back
  // of what a kernel stored, profiles/r02_notes.md: the chain saves two boundaries per layer) it is 3.86 vs 3.97 ms/step.
  c->fuse_ffn = !(getenv("AMT_FUSE_FFN") && atoi(getenv("AMT_FUSE_FFN")) == 0);
  c->prerot = !(getenv("AMT_PREROT") && atoi(getenv("AMT_PREROT")) == 0);
  c->opt_persist = !(getenv("AMT_OPT_PERSIST") && atoi(getenv("AMT_OPT_PERSIST")) == 0);
  c->persist_grid_extra = getenv("AMT_OPT_PERSIST_EXTRA") ? atoi(getenv("AMT_OPT_PERSIST_EXTRA")) : 0;
  c->persist_spins = getenv("AMT_OPT_PERSIST_SPINS") ? (unsigned)atol(getenv("AMT_OPT_PERSIST_SPINS")) : 0u;
  if (const char* d = getenv("AMT_OPT_PERSIST_DELAY")) {
    int wg = -1; unsigned polls = 0;
    if (sscanf(d, "%d:%u", &wg, &polls) == 2) { c->persist_delay_wg = wg; c->persist_delay_spins = polls; }
  }
  { hipDeviceProp_t pr; c->n_cus = (hipGetDeviceProperties(&pr, c->device) == hipSuccess) ? pr.multiProcessorCount : 0; }
  // XCD-team launches (team.h): opt-in while they are not faster than one launch per Block (profiles/r02_notes.md)
  c->team_cnn = getenv("AMT_TEAM") && atoi(getenv("AMT_TEAM")) != 0;
  // the transformer stack's forward as ONE team launch (tlayer.hip)
  c->tlayer = (getenv("AMT_TLAYER") && atoi(getenv("AMT_TLAYER")) != 0) && tlayer_fwd_device_ok();   // opt-in while it is slower than the chains
  if (c->team_cnn || c->tlayer) {
    c->team = team_state_create();
    if (c->team == nullptr) { g_create_error = "hipMalloc(team state)"; delete c; return 1; }
  }
  if (c->tlayer && getenv("AMT_TLAYER_STAMPS") && atoi(getenv("AMT_TLAYER_STAMPS")) != 0) {
    if (hipMalloc(reinterpret_cast<void**>(&c->tl_stamps), sizeof(unsigned long long) * 256 * TLAYER_MAX * 32) != hipSuccess) c->tl_stamps = nullptr;
    else (void)hipMemset(c->tl_stamps, 0, sizeof(unsigned long long) * 256 * TLAYER_MAX * 32);
  }
  c->cnn_late_env = getenv("AMT_CNN_WGRAD_LATE") != nullptr;
  c->cnn_late_max_c = c->cnn_late_env ? atoi(getenv("AMT_CNN_WGRAD_LATE")) : 0;
  if (c->cnn_late_max_c == 1) c->cnn_late_max_c = 1 << 20;
  if (c->nochain == 0) {   // keep only the conversions the selected execution mode reads (debug mask: keep all)
    const bool chain_tr = c->fuse && cfg->dims[cfg->n_stages - 1] == 256 && cfg->ffn_inter == 512 &&
                          cfg->n_heads * cfg->head_dim == 256 && cfg->kv_compress == 64;
    std::vector<ConvDesc> kept;
    for (size_t i = 0; i < c->L.conv.size(); ++i) {
      const int need = c->L.conv_need[i];
      const bool use = need == NEED_ALWAYS || (need == NEED_UNFUSED_BLOCK && !c->fuse) || (need == NEED_FUSED_BLOCK && c->fuse) ||
                       (need == NEED_UNFUSED_TR && !chain_tr) || (need == NEED_FUSED_TR && chain_tr) ||
                       (need == NEED_FFN_CHAIN && chain_tr && c->fuse_ffn) || (need == NEED_FFN_TILED && !(chain_tr && c->fuse_ffn));
      if (use) kept.push_back(c->L.conv[i]);
    }
    c->L.conv.swap(kept);
  }
  const Layout& L = c->L;
  auto fail = [&](const char* what, hipError_t err) {
    g_create_error = std::string(what) + ": " + hipGetErrorString(err);
    amt_destroy(c);
    return 1;
  };
  if ((he = hipMalloc(&c->params, L.n_params * sizeof(float))) != hipSuccess) return fail("hipMalloc(params)", he);
  if ((he = hipMemset(c->params, 0, L.n_params * sizeof(float))) != hipSuccess) return fail("hipMemset", he);
  if ((he = hipMalloc(&c->wc, L.n_compute * c->esz)) != hipSuccess) return fail("hipMalloc(compute copies)", he);
  if ((he = hipMemset(c->wc, 0, L.n_compute * c->esz)) != hipSuccess) return fail("hipMemset", he);   // packed copies: padding stays zero
  if ((he = hipMalloc(&c->head_bias, L.VP * sizeof(float))) != hipSuccess) return fail("hipMalloc(head bias)", he);
  if ((he = hipMalloc(&c->loss_acc, 2 * sizeof(float))) != hipSuccess) return fail("hipMalloc(loss)", he);
  if ((he = hipMalloc(reinterpret_cast<void**>(&c->seed_word), sizeof(uint64_t))) != hipSuccess) return fail("hipMalloc(seed word)", he);
  if ((he = hipMemset(c->seed_word, 0, sizeof(uint64_t))) != hipSuccess) return fail("hipMemset", he);
  // The accumulator is bucket-shaped (amt_grad_bucket): [.. | 4 control floats | n gradients | 2 control floats | ..], so that a
  // data-parallel host can run its collectives ON it (no copy into a bucket of its own).  64 floats of room in front of the
  // shift that aligns layer 0, 64 behind the gradients.
  if ((he = hipMalloc(reinterpret_cast<void**>(&c->gacc), (L.n_params + 192) * sizeof(float))) != hipSuccess)
    return fail("hipMalloc(gradient accumulator)", he);
  if ((he = hipMemset(c->gacc, 0, (L.n_params + 192) * sizeof(float))) != hipSuccess) return fail("hipMemset", he);
  c->gacc_pad = 64;
  if (!L.layers.empty()) {
    const LayerP& l0 = L.layers[0];
    size_t a0 = l0.ln1_w;
    for (size_t o : {l0.ln1_b, l0.wqc, l0.wkv, l0.wo, l0.ln2_w, l0.ln2_b, l0.w1, l0.b1, l0.w2, l0.b2}) a0 = o < a0 ? o : a0;
    c->gacc_pad = 64 + (int)((64 - a0 % 64) % 64);
  }
  {
    std::vector<GammaDesc> gd;
    int row0 = 0;
    for (const StageP& st : L.stages)
      for (const BlockP& b : st.blocks) {
        gd.push_back(GammaDesc{b.pw2_w, b.pw2_b, b.gamma, st.C, row0});
        row0 += st.C;
      }
    c->n_gamma_desc = (int)gd.size();
    c->gamma_rows = row0;
    if (!gd.empty()) {
      if ((he = hipMalloc(reinterpret_cast<void**>(&c->gamma_desc_dev), gd.size() * sizeof(GammaDesc))) != hipSuccess) return fail("hipMalloc", he);
      (void)hipMemcpy(c->gamma_desc_dev, gd.data(), gd.size() * sizeof(GammaDesc), hipMemcpyHostToDevice);
    }
  }
  std::vector<int> prefix(L.conv.size() + 1, 0);
  for (size_t i = 0; i < L.conv.size(); ++i)
    prefix[i + 1] = prefix[i] + ((L.conv[i].rows + 31) / 32) * ((L.conv[i].cols + 63) / 64);   // head.hip: convert_batch_kernel tiles
  c->conv_tiles = prefix.back();
  if ((he = hipMalloc(&c->conv_dev, L.conv.size() * sizeof(ConvDesc))) != hipSuccess) return fail("hipMalloc", he);
  if ((he = hipMalloc(&c->conv_prefix_dev, prefix.size() * sizeof(int))) != hipSuccess) return fail("hipMalloc", he);
  (void)hipMemcpy(c->conv_dev, L.conv.data(), L.conv.size() * sizeof(ConvDesc), hipMemcpyHostToDevice);
  (void)hipMemcpy(c->conv_prefix_dev, prefix.data(), prefix.size() * sizeof(int), hipMemcpyHostToDevice);
  *out = c;
  return 0;
}

void amt_destroy(amt_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  void* ptrs[] = {c->params, c->wc, c->head_bias, c->conv_dev, c->conv_prefix_dev, c->rope_cos, c->rope_sin,
                  c->arena.base, c->adam_m, c->adam_v, c->adam_upd, c->lr_scale, c->norm2, c->loss_acc,
                  c->partials, c->amax, c->gacc, c->team, c->seed_word, c->lr_idx, c->lr_tab, c->half_sig, c->gamma_desc_dev, c->tl_stamps};
  if (c->half_host) { (void)hipDeviceSynchronize(); (void)hipHostFree(const_cast<unsigned*>(c->half_host)); }
  for (const amt_ctx::PlanRec& r : c->plans)
    if (r.desc_dev) (void)hipFree(r.desc_dev);
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (c->persist_fail) (void)hipHostFree(c->persist_fail);
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  if (c->side) (void)hipStreamDestroy(c->side);
  delete c;
}

int32_t amt_load_weights(amt_ctx* c, const float* flat, uint64_t n, int32_t on_device, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n != c->L.n_params) { c->err = "amt_load_weights: wrong parameter count"; return 1; }
  HIPCHK(c, hipMemcpyAsync(c->params, flat, n * sizeof(float), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                           s));
  return refresh(c, s);
}

float* amt_weights_ptr(amt_ctx* c) { return c->params; }

int32_t amt_refresh_weights(amt_ctx* c, void* stream) { return refresh(c, static_cast<hipStream_t>(stream)); }

int32_t amt_set_rope(amt_ctx* c, const float* cos_table, const float* sin_table, int32_t n_pos) {
  if (n_pos <= 0) { c->err = "amt_set_rope: n_pos must be positive"; return 1; }
  c->rope_user = true;
  return upload_rope(c, cos_table, sin_table, n_pos);
}

int32_t amt_set_attn_fp8(amt_ctx* c, int32_t enable) {
  if (enable && c->dt != DT_BF16) { c->err = "amt_set_attn_fp8: fp8 attention needs a bf16 context"; return 1; }
  if (enable && c->amax == nullptr) {
    const size_t bytes = c->L.layers.size() * 8 * sizeof(float);
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->amax), bytes));
    HIPCHK(c, hipMemset(c->amax, 0, bytes));
  }
  c->attn_fp8 = enable != 0;
  return 0;
}

int32_t amt_forward(amt_ctx* c, const void* audio, int32_t B, int32_t T, float* logits, float* probs, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (check_shape(c, B, T)) return 1;
  if (ensure_rope(c, T / 320)) return 1;
  if (plan(c, B, T, 0, s)) return 1;
  c->arena.off = 0;
  if (run_forward(c, s, audio, B, T, false, false, 0)) return 1;
  if (c->arena.overflow) { c->err = "internal error: workspace plan too small"; return 1; }
  head_probs(s, c->logits_pad, c->L.VP, logits, probs, B * (T / 320), c->cfg.n_events);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { c->err = std::string("amt_forward: ") + hipGetErrorString(e); return 1; }
  return 0;
}

int32_t amt_forward_train(amt_ctx* c, const void* audio, int32_t B, int32_t T, uint64_t seed, float* logits, float* probs,
                          void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (check_shape(c, B, T)) return 1;
  if (ensure_rope(c, T / 320)) return 1;
  if (plan(c, B, T, 1, s)) return 1;           // the training plan: the forward of the training step, dropout on
  c->arena.off = 0;
  if (run_forward(c, s, audio, B, T, true, true, seed)) return 1;
  if (c->arena.overflow) { c->err = "internal error: workspace plan too small"; return 1; }
  head_probs(s, c->logits_pad, c->L.VP, logits, probs, B * (T / 320), c->cfg.n_events);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { c->err = std::string("amt_forward_train: ") + hipGetErrorString(e); return 1; }
  return 0;
}

int32_t amt_forward_backward_accum(amt_ctx* c, const void* audio, const void* labels, int32_t B, int32_t T, float scale,
                                   uint64_t seed, int32_t train, float* loss_out, float* grads_flat, int32_t* grads_valid,
                                   float* logits, int32_t phase, int32_t accum_index, int32_t accum_count, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (phase < 0 || phase > 3) { c->err = "phase must be 0 (whole step), 1, 2 or 3 (whole step + first-half signal)"; return 1; }
  if (phase == 3 && c->use_side) { c->err = "phase 3 (first-half signal) is not available with AMT_SIDE_STREAM"; return 1; }
  if (phase == 3 && grads_flat != nullptr) { c->err = "phase 3 (first-half signal) belongs to the fused protocol: grads_flat must be NULL"; return 1; }
  if (phase == 3 && c->half_sig == nullptr) {
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->half_sig), 8));
    HIPCHK(c, hipMemset(c->half_sig, 0, 8));
    unsigned* hw = nullptr;
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&hw), sizeof(unsigned), hipHostMallocMapped));
    *hw = 0u;
    c->half_host = hw;
    HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void**>(&c->half_host_dev), hw, 0));
    HIPCHK(c, hipDeviceSynchronize());
  }
  if (accum_count < 1 || accum_index < 0 || accum_index >= accum_count) { c->err = "accum_index must be in [0, accum_count)"; return 1; }
  if (check_shape(c, B, T)) return 1;
  if (!(scale > 0.f)) { c->err = "scale must be positive"; return 1; }
  if (phase != 2) {
    if (ensure_rope(c, T / 320)) return 1;
    if (plan(c, B, T, 1, s)) return 1;
    c->arena.off = 0;
    if (run_forward(c, s, audio, B, T, true, train != 0, seed)) return 1;
    c->fwd_end_off = c->arena.off;
    if (logits) head_probs(s, c->logits_pad, c->L.VP, logits, nullptr, B * (T / 320), c->cfg.n_events);
  } else {
    if (c->plan_B != B || c->plan_T != T || c->plan_train != 1) { c->err = "phase 2 without a matching phase 1"; return 1; }
    c->arena.off = c->fwd_end_off;
  }
  if (run_backward(c, s, audio, labels, B, T, scale, train != 0, seed, grads_flat, grads_valid, phase, accum_index, accum_count))
    return 1;
  if (c->arena.overflow) { c->err = "internal error: workspace plan too small"; return 1; }
  // the loss accumulators hold the SUM over the minibatches so far of each minibatch's mean loss (train.py:60-62, 292)
  if (phase != 2 && loss_out != nullptr && accum_index == accum_count - 1) scale_copy_f32(s, c->loss_acc, loss_out, 2, 1.f / (float)accum_count);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { c->err = std::string("amt_forward_backward: ") + hipGetErrorString(e); return 1; }
  return 0;
}

int32_t amt_forward_backward(amt_ctx* c, const void* audio, const void* labels, int32_t B, int32_t T, float scale,
                             uint64_t seed, int32_t train, float* loss_out, float* grads_flat, int32_t* grads_valid,
                             float* logits, void* stream) {
  return amt_forward_backward_accum(c, audio, labels, B, T, scale, seed, train, loss_out, grads_flat, grads_valid, logits, 0, 0, 1,
                                    stream);
}

int32_t amt_forward_backward_phase(amt_ctx* c, const void* audio, const void* labels, int32_t B, int32_t T, float scale,
                                   uint64_t seed, int32_t train, float* loss_out, float* grads_flat, int32_t* grads_valid,
                                   float* logits, int32_t phase, void* stream) {
  if (phase != 1 && phase != 2) { c->err = "phase must be 1 or 2"; return 1; }
  return amt_forward_backward_accum(c, audio, labels, B, T, scale, seed, train, loss_out, grads_flat, grads_valid, logits, phase, 0,
                                    1, stream);
}

uint64_t amt_grad_bucket_split(const amt_ctx* c) { return c->L.fn_w; }

uint64_t amt_workspace_bytes(const amt_ctx* c) { return c->arena.cap; }

uint64_t amt_plan_generation(const amt_ctx* c) { return c->plan_gen; }

uint64_t* amt_seed_word(amt_ctx* c) { return c->seed_word; }

int32_t amt_pack_tail(amt_ctx* c, const float* loss, const int32_t* grads_valid, float* head_valid, float* tail_loss,
                      float* tail_valid, void* stream) {
  if (loss == nullptr || grads_valid == nullptr) { c->err = "amt_pack_tail: null loss / grads_valid"; return 1; }
  pack_tail(static_cast<hipStream_t>(stream), loss, grads_valid, head_valid, tail_loss, tail_valid);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { c->err = std::string("amt_pack_tail: ") + hipGetErrorString(e); return 1; }
  return 0;
}

int32_t amt_advance_seed(amt_ctx* c, uint64_t delta, void* stream) {
  hipLaunchKernelGGL(advance_seed_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), c->seed_word, delta);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { c->err = std::string("amt_advance_seed: ") + hipGetErrorString(e); return 1; }
  return 0;
}

int32_t amt_set_label_dtype(amt_ctx* c, int32_t dtype) {
  if (dtype != AMT_F32 && dtype != AMT_F16) { c->err = "amt_set_label_dtype: AMT_F32 or AMT_F16"; return 1; }
  c->labels_f16 = dtype == AMT_F16;
  return 0;
}

int32_t amt_tlayer_active(const amt_ctx* c) { return c->tlayer && c->team != nullptr && !c->team_dead ? 1 : 0; }

int32_t amt_tlayer_stamps(const amt_ctx* c, uint64_t* out, int32_t max_words) {
  if (c->tl_stamps == nullptr || out == nullptr) return 0;
  const int n = std::min<int>(max_words, 256 * TLAYER_MAX * 32);
  if (hipMemcpy(out, c->tl_stamps, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return n;
}

int32_t amt_team_status(const amt_ctx* c) {
  const int st = c->team ? team_state_timeout(c->team) : 0;
  if (st != 0) c->team_dead = true;     // sticky on the host as well: the CNN stages go back to one launch per Block
  return st;
}

void amt_profile_begin(void) {
  Profiler& p = profiler();
  p.recs.clear();
  p.used = 0;
  p.on = true;
}

int32_t amt_profile_end(amt_prof_entry* out, int32_t max_entries) {
  Profiler& p = profiler();
  p.on = false;
  (void)hipDeviceSynchronize();
  std::vector<amt_prof_entry> agg(p.names.size());
  for (size_t i = 0; i < p.names.size(); ++i) {
    memset(&agg[i], 0, sizeof(amt_prof_entry));
    snprintf(agg[i].name, sizeof(agg[i].name), "%s", p.names[i].c_str());
  }
  for (const ProfRec& r : p.recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
    agg[r.cls].launches += 1;
    agg[r.cls].total_ms += ms;
    agg[r.cls].flops += r.flops;
    agg[r.cls].bytes += r.bytes;
  }
  int32_t n = 0;
  for (const amt_prof_entry& e : agg)
    if (e.launches > 0 && n < max_entries) out[n++] = e;
  p.recs.clear();
  p.used = 0;
  return n;
}

int32_t amt_optimizer_init(amt_ctx* c, float layer_lr_decay) {
  const Layout& L = c->L;
  const size_t n = L.n_params;
  if (!c->adam_m) {
    HIPCHK(c, hipMalloc(&c->adam_m, n * sizeof(float)));
    HIPCHK(c, hipMalloc(&c->adam_v, n * sizeof(float)));
    HIPCHK(c, hipMalloc(&c->adam_upd, n * sizeof(float)));
    HIPCHK(c, hipMalloc(&c->lr_scale, n * sizeof(float)));
    HIPCHK(c, hipMalloc(&c->lr_idx, (n + 3) / 4 * 4));
    HIPCHK(c, hipMalloc(&c->lr_tab, 256 * sizeof(float)));
    if (hipHostMalloc(reinterpret_cast<void**>(&c->persist_fail), sizeof(int), hipHostMallocMapped) == hipSuccess &&
        hipHostGetDevicePointer(reinterpret_cast<void**>(&c->persist_fail_dev), c->persist_fail, 0) == hipSuccess) {
      *c->persist_fail = 0;
    } else {                                                                                 // no mapped word: two launches
      c->persist_fail = c->persist_fail_dev = nullptr; (void)hipGetLastError(); c->opt_persist = false;
    }
    // [sum of squares, reader count]: adamw_apply leaves both zero; [2]: int count of applied updates; [4..7]: grid barrier of the
    // one-launch fused form (arrivals, launches, sticky timeout, decision word); [8 ..): one partial-sum slot per workgroup of
    // the fused forms' norm pass (head.hip: sum_partials)
    HIPCHK(c, hipMalloc(&c->norm2, (8 + 4096) * sizeof(float)));
    HIPCHK(c, hipMemset(c->norm2, 0, (8 + 4096) * sizeof(float)));
    HIPCHK(c, hipMemset(c->adam_m, 0, n * sizeof(float)));
    HIPCHK(c, hipMemset(c->adam_v, 0, n * sizeof(float)));
  }
  if (c->lr_scale_decay != layer_lr_decay) {
    int max_depth = 0;
    for (const Seg& sg : L.segs) if (sg.lr_depth > max_depth) max_depth = sg.lr_depth;
    std::vector<float> sc(n, 1.f);
    // the same factors as a byte index into a table (entry 0: 1.0; entry d + 1: the factor of depth d)
    std::vector<unsigned char> idx((n + 3) / 4 * 4, 0);
    std::vector<float> tab(256, 1.f);
    const bool idx_ok = max_depth < 255;
    for (const Seg& sg : L.segs) {
      float f = sg.lr_depth >= 0 ? powf(layer_lr_decay, (float)(max_depth - sg.lr_depth)) : 1.f;
      for (size_t i = 0; i < (size_t)sg.rows * sg.cols; ++i) sc[sg.off + i] = f;
      if (idx_ok && sg.lr_depth >= 0) {
        tab[sg.lr_depth + 1] = f;
        for (size_t i = 0; i < (size_t)sg.rows * sg.cols; ++i) idx[sg.off + i] = (unsigned char)(sg.lr_depth + 1);
      }
    }
    HIPCHK(c, hipMemcpy(c->lr_scale, sc.data(), n * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->lr_idx, idx.data(), idx.size(), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->lr_tab, tab.data(), 256 * sizeof(float), hipMemcpyHostToDevice));
    c->lr_idx_ok = idx_ok;
    c->lr_scale_decay = layer_lr_decay;
  }
  HIPCHK(c, hipDeviceSynchronize());
  return 0;
}

namespace {
// the fused protocol's optimizer (include/amt.h): norm pass + commit pass over the context's accumulator
int adamw_from_accumulator(amt_ctx* c, hipStream_t s, float lr_base, float b1, float b2, float eps, float weight_decay,
                           float clip_norm, int step, float grad_mul, const float* loss_dev, const float* valid_f32,
                           int32_t* valid_out) {
  const size_t n = c->L.n_params;
  int* count = reinterpret_cast<int*>(c->norm2 + 2);
  if (!c->fused_used) { c->err = "amt_adamw_step(grads_flat = NULL): no backward pass has left gradients in the context (call amt_forward_backward* with grads_flat = NULL first)"; return 1; }
  if (c->opt_persist && c->persist_fail != nullptr && *reinterpret_cast<volatile int*>(c->persist_fail) != 0) {
    c->opt_persist = false;         // its grid barrier did not open (workgroups not co-resident; the step was still applied, head.hip): two launches from now on
    c->persist_timeouts++;
    *reinterpret_cast<volatile int*>(c->persist_fail) = 0;
    c->err = "the one-launch fused optimizer timed out at its grid barrier (one workgroup committed that step alone: nothing was lost); using the two-launch form";
  }
  // A captured launch would replay with no host in between to see a time-out and switch forms (each replay would wait for
  // its short spin limit and commit through one workgroup): under stream capture the step takes the two launches.
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &cs);
  const bool persist = c->opt_persist && c->lr_idx_ok && cs == hipStreamCaptureStatusNone;
  adamw_fused(s, c->params, c->gacc + c->gacc_pad, c->adam_m, c->adam_v, c->lr_scale, n, c->gacc_mul * grad_mul, lr_base, b1, b2,
              eps, weight_decay, step, count, c->norm2, clip_norm, nullptr, valid_f32, loss_dev ? loss_dev : c->loss_acc, valid_out,
              c->n_cus + c->persist_grid_extra, persist, c->lr_idx, c->lr_tab, c->persist_fail_dev, c->persist_spins,
              c->persist_delay_wg, c->persist_delay_spins);
  return refresh(c, s);
}
}  // namespace

int32_t amt_adamw_step_gated(amt_ctx* c, const float* grads_flat, float lr_base, float layer_lr_decay, float b1, float b2,
                             float eps, float weight_decay, float clip_norm, int32_t step, const int32_t* valid_i32,
                             const float* valid_f32, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t n = c->L.n_params;
  // optax counts from 1: step <= 0 makes the bias corrections 1 - b^step zero or negative (m / 0 = inf in every parameter)
  if (step < 1 && step != -1) { c->err = "amt_adamw_step: step must be >= 1 (the first optimizer step is step 1), or -1 for the device count"; return 1; }
  if (!c->adam_m || c->lr_scale_decay != layer_lr_decay) {
    if (amt_optimizer_init(c, layer_lr_decay)) return 1;
  }
  int* count = reinterpret_cast<int*>(c->norm2 + 2);
  if (grads_flat == nullptr)
    // fused protocol: the gradients are the sums the last amt_forward_backward*(grads_flat = NULL) left in the context's
    // accumulator; valid_i32 is an OUTPUT here (the all-finite flag of this step, 1 = update applied)
    return adamw_from_accumulator(c, s, lr_base, b1, b2, eps, weight_decay, clip_norm, step, 1.f, nullptr, valid_f32,
                                  const_cast<int32_t*>(valid_i32));
  adamw_compute(s, c->params, grads_flat, c->adam_m, c->adam_v, c->adam_upd, c->lr_scale, n, lr_base, b1, b2, eps,
                weight_decay, step, count, c->norm2, valid_i32, valid_f32);
  adamw_apply(s, c->params, c->adam_upd, n, c->norm2, clip_norm, valid_i32, valid_f32, step, count);
  return refresh(c, s);
}

int32_t amt_adamw_step_reduced(amt_ctx* c, float lr_base, float layer_lr_decay, float b1, float b2, float eps,
                               float weight_decay, float clip_norm, int32_t step, float grad_mul, const float* loss_sum,
                               int32_t* valid_out, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (step < 1 && step != -1) { c->err = "amt_adamw_step: step must be >= 1 (the first optimizer step is step 1), or -1 for the device count"; return 1; }
  if (!(grad_mul > 0.f)) { c->err = "amt_adamw_step_reduced: grad_mul must be positive (1 / world size)"; return 1; }
  if (!c->adam_m || c->lr_scale_decay != layer_lr_decay) {
    if (amt_optimizer_init(c, layer_lr_decay)) return 1;
  }
  return adamw_from_accumulator(c, s, lr_base, b1, b2, eps, weight_decay, clip_norm, step, grad_mul, loss_sum, nullptr, valid_out);
}

float* amt_grad_bucket(amt_ctx* c, uint64_t* head, uint64_t* n, uint64_t* tail) {
  if (head) *head = 4;
  if (n) *n = c->L.n_params;
  if (tail) *tail = 2;
  return c->gacc + c->gacc_pad - 4;
}

int32_t amt_first_half_wait(amt_ctx* c, int32_t timeout_ms) {
  if (c->half_host == nullptr) { c->err = "amt_first_half_wait: no phase-3 step has been issued on this context"; return 1; }
  const uint32_t target = c->half_base + ++c->half_waits;      // the k-th wait since the last sync consumes the k-th signal
  const auto t0 = std::chrono::steady_clock::now();
  unsigned polls = 0;
  while ((int32_t)(*c->half_host - target) < 0) {
    if ((++polls & 1023u) == 0 &&
        std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > (timeout_ms > 0 ? timeout_ms : 10000)) {
      c->half_timeouts++;
      c->err = "amt_first_half_wait: the step did not publish its first-half signal in time";
      return 1;
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return 0;
}

int32_t amt_first_half_sync(amt_ctx* c, void* stream) {
  if (c->half_host == nullptr) return 0;
  HIPCHK(c, hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  c->half_base = *c->half_host;
  c->half_waits = 0;
  return 0;
}

int32_t amt_first_half_status(amt_ctx* c) { return c->half_timeouts; }

int32_t amt_optimizer_status(amt_ctx* c) {
  if (c->persist_fail != nullptr && *reinterpret_cast<volatile int*>(c->persist_fail) != 0) {
    c->persist_timeouts++;
    *reinterpret_cast<volatile int*>(c->persist_fail) = 0;      // counted; the device's sticky word keeps later launches of the one-launch form dead
    if (c->opt_persist) {
      c->opt_persist = false;
      c->err = "the one-launch fused optimizer timed out at its grid barrier (one workgroup committed that step alone: nothing was lost); using the two-launch form";
    }
  }
  return c->persist_timeouts;
}

int32_t amt_grad_accumulator_reset(amt_ctx* c, void* stream) {
  fill_f32(static_cast<hipStream_t>(stream), c->gacc + c->gacc_pad, 0.f, c->L.n_params);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { c->err = std::string("amt_grad_accumulator_reset: ") + hipGetErrorString(e); return 1; }
  return 0;
}

float* amt_optimizer_moment_ptr(amt_ctx* c, int32_t which) { return which == 0 ? c->adam_m : (which == 1 ? c->adam_v : nullptr); }

int32_t amt_optimizer_set_count(amt_ctx* c, int32_t count) {
  if (!c->norm2) { c->err = "amt_optimizer_set_count: call amt_optimizer_init first"; return 1; }
  if (count < 0) { c->err = "amt_optimizer_set_count: count must be >= 0"; return 1; }
  HIPCHK(c, hipDeviceSynchronize());
  HIPCHK(c, hipMemcpy(c->norm2 + 2, &count, sizeof(int), hipMemcpyHostToDevice));
  return 0;
}

int32_t amt_optimizer_count(amt_ctx* c) {
  if (!c->norm2) return 0;
  int w[5] = {0, 0, 0, 0, 0};     // [0] count, [4] sticky timeout word of the one-launch fused form (head.hip)
  if (hipMemcpy(w, c->norm2 + 2, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (w[4] != 0) (void)amt_optimizer_status(c);      // the sticky word: switch to the two launches, count the time-out
  return w[0];
}

// ---- residual-stream checkpoints of the last training-mode forward (include/amt_ops.h; probes/bf16_layer_probe.py)
namespace {
struct StreamCk { std::string name; void* ptr; long rows; int cols; };
std::vector<StreamCk> stream_checkpoints(amt_ctx* c) {
  std::vector<StreamCk> v;
  if (c->plan_train != 1 || c->ssave.empty() || c->plan_B <= 0) return v;
  const int B = c->plan_B;
  int Lcur = c->plan_T / 5;
  for (size_t si = 0; si < c->ssave.size(); ++si) {
    if (si > 0) Lcur /= 2;
    const int C = c->L.stages[si].C;
    for (size_t j = 0; j < c->ssave[si].blocks.size(); ++j)
      v.push_back(StreamCk{"cnn." + std::to_string(si) + "." + std::to_string(j + 1) + ".x", c->ssave[si].blocks[j].x, (long)B * Lcur, C});
  }
  const long N = (long)B * (c->plan_T / 320);
  v.push_back(StreamCk{"cnn.norm.x", c->fn_x, N, c->L.H});
  for (size_t li = 0; li < c->lsave.size(); ++li)
    v.push_back(StreamCk{"tl." + std::to_string(li) + ".x", c->lsave[li].x, N, c->L.H});
  v.push_back(StreamCk{"head.x", c->hd_x, N, c->L.H});
  return v;
}
}  // namespace

int32_t amt_op_stream_count(amt_ctx* c) { return (int32_t)stream_checkpoints(c).size(); }

int32_t amt_op_stream_checkpoint(amt_ctx* c, int32_t index, char name[64], void** ptr, int64_t* rows, int32_t* cols) {
  const std::vector<StreamCk> v = stream_checkpoints(c);
  if (index < 0 || index >= (int32_t)v.size() || v[index].ptr == nullptr) { c->err = "amt_op_stream_checkpoint: bad index or no saved training forward"; return 1; }
  snprintf(name, 64, "%s", v[index].name.c_str());
  *ptr = v[index].ptr; *rows = v[index].rows; *cols = v[index].cols;
  return 0;
}

int32_t amt_adamw_step(amt_ctx* c, const float* grads_flat, float lr_base, float layer_lr_decay, float b1, float b2,
                       float eps, float weight_decay, float clip_norm, int32_t step, void* stream) {
  return amt_adamw_step_gated(c, grads_flat, lr_base, layer_lr_decay, b1, b2, eps, weight_decay, clip_norm, step, nullptr,
                              nullptr, stream);
}

}  // extern "C"
