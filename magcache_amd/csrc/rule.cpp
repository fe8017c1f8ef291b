// Host-side MagCache decision rule: a literal restatement, in double precision, of the scalar
// state machine every reference script carries
//   Wan2.1      MagCache4Wan2.1/magcache_generate.py:277-292, :306-311
//   Wan2.2      MagCache4Wan2.2/magcache_generate.py:290-317, :328-333
//   HunyuanVideo MagCache4HunyuanVideo/magcache_sample_video.py:88-102
//   FLUX        MagCache4FLUX/magcache_flux.py:326-338, :432-437   (FLUX-Kontext: magcache_flux_kontext.py:329-340)
//   FramePack   MagCache4FramePack/magcache_demo_gradio.py:253-271, :298-300
//   OmniGen2    MagCache4OmniGen2/magcache/magcache_utils.py:343-356, :368-376 (accumulated_steps starts at 3, :44)
//   Qwen-Image  MagCache4QwenImage/magcache_generate.py:205-219, :242-244 (no accumulator reset at wrap-around)
//   eval Wan    eval/magcache/experiments/Wan2.1_EVAL/wan_magcache.py:770-787, :807-815
//   eval Open-Sora  eval/magcache/experiments/opensora.py:297-309, :348-354
// It never touches the device (the reference's decision has no .item()/sync either), so the launch
// path stays asynchronous.  Python floats are IEEE doubles: the arithmetic below is bit-identical.
#include <cmath>
#include <vector>

#include "../../include/magcache_hip.h"

struct mc_rule {
  int variant, num_steps, K, split_step;
  double thresh, retention;
  std::vector<double> ratios;
  int cnt = 0;
  double acc_err[2] = {0.0, 0.0};
  int acc_steps[2] = {0, 0};
  double acc_ratio[2] = {1.0, 1.0};
};

namespace {

bool two_slot(int variant) {
  return variant == MC_RULE_WAN21 || variant == MC_RULE_WAN22_T2V || variant == MC_RULE_WAN22_I2V ||
         variant == MC_RULE_WAN22_TI2V || variant == MC_RULE_QWEN || variant == MC_RULE_EVAL_WAN;
}

// comparison of the accumulated error with the threshold: '<' in the Wan-style scripts, '<=' everywhere else
bool strict_less(int variant) {
  return variant == MC_RULE_WAN21 || variant == MC_RULE_WAN22_T2V || variant == MC_RULE_WAN22_I2V ||
         variant == MC_RULE_WAN22_TI2V || variant == MC_RULE_QWEN;
}

// index of the table entry the call at counter cnt reads.  May be negative for the two eval variants when num_steps
// is short (eval Wan opens its gate at int(n*0.2) but reads ratio[t-10]): the reference indexes a Python list / ndarray,
// which wraps a negative index once (ratios[-3] = third from the end) and raises beyond that -- table_entry() below
// does the same wrap, and mc_rule_create rejects every configuration that would raise.
int ratio_index(const mc_rule* r) {
  if (r->variant == MC_RULE_EVAL_WAN) return r->cnt - 10;       // "ratios are cached after 10 steps"
  if (r->variant == MC_RULE_EVAL_OPENSORA) return r->cnt - 1;
  return r->cnt;
}

bool index_valid(long idx, long n) { return idx >= -n && idx < n; }

double table_entry(const mc_rule* r, int idx) {
  const long n = (long)r->ratios.size();
  return r->ratios[(size_t)(idx < 0 ? idx + n : idx)];
}

bool gate_open(const mc_rule* r) {
  const int n = r->num_steps, cnt = r->cnt, sp = r->split_step;
  const double R = r->retention;
  switch (r->variant) {
    case MC_RULE_WAN21:
    case MC_RULE_QWEN:
    case MC_RULE_WAN22_TI2V: return cnt >= (int)(n * R);
    case MC_RULE_FRAMEPACK: return cnt >= (int)(R * n) && cnt >= 1;
    case MC_RULE_OMNIGEN2: return cnt >= (int)std::ceil(R * n);
    case MC_RULE_EVAL_WAN: return cnt >= (int)(n * 0.2);           // skip_time is hard-wired to 20 % (:772)
    case MC_RULE_EVAL_OPENSORA: return cnt >= (int)(R * n);        // skip_time = 6 = 30 * 0.2 (:422)
    case MC_RULE_HUNYUAN: return cnt >= (int)(R * n);
    case MC_RULE_FLUX: return cnt >= (int)(R * n + 0.5);
    case MC_RULE_WAN22_I2V: return !(cnt < (int)(sp + (n - sp) * R));
    case MC_RULE_WAN22_T2V:
      return !(cnt < (int)(sp * R) || ((double)cnt <= ((n - sp) * R + sp) && cnt >= sp));
    default: return false;
  }
}

}  // namespace

extern "C" {

mc_rule* mc_rule_create(int variant, int num_steps, double thresh, int K, double retention_ratio,
                        const double* mag_ratios, int n_ratios, int split_step) {
  if (variant < MC_RULE_WAN21 || variant > MC_RULE_EVAL_OPENSORA || num_steps <= 0 || !mag_ratios || n_ratios <= 0)
    return nullptr;
  mc_rule* r = new mc_rule();
  r->variant = variant;
  r->num_steps = num_steps;
  r->thresh = thresh;
  r->K = K;
  r->retention = retention_ratio;
  r->split_step = split_step;
  r->ratios.assign(mag_ratios, mag_ratios + n_ratios);
  if (variant == MC_RULE_OMNIGEN2) r->acc_steps[0] = 3;   // MagCacheParams.accumulated_steps: int = 3 (:44)
  // every call whose gate is open reads one table entry: all of them must exist (Python would raise IndexError)
  for (r->cnt = 0; r->cnt < num_steps; ++r->cnt) {
    if (gate_open(r) && !index_valid(ratio_index(r), n_ratios)) {
      delete r;
      return nullptr;
    }
  }
  r->cnt = 0;
  return r;
}

void mc_rule_destroy(mc_rule* r) { delete r; }

int mc_rule_step(mc_rule* r, int* branch) {
  const int p = two_slot(r->variant) ? (r->cnt % 2) : 0;
  if (branch) *branch = p;
  bool skip = false;
  if (r->variant == MC_RULE_FRAMEPACK && r->cnt == 0) {  // "initialize MagCache" at the first call of a section
    r->acc_ratio[0] = 1.0;
    r->acc_steps[0] = 0;
    r->acc_err[0] = 0.0;
  }
  if (gate_open(r)) {
    const double cur = table_entry(r, ratio_index(r));
    r->acc_ratio[p] = r->acc_ratio[p] * cur;
    r->acc_steps[p] += 1;
    // Open-Sora's eval script accumulates the SIGNED distance (no abs, opensora.py:301)
    r->acc_err[p] += r->variant == MC_RULE_EVAL_OPENSORA ? 1.0 - r->acc_ratio[p] : std::fabs(1.0 - r->acc_ratio[p]);
    bool ok;
    if (strict_less(r->variant)) {
      ok = r->acc_err[p] < r->thresh && r->acc_steps[p] <= r->K;
    } else {
      ok = r->acc_err[p] <= r->thresh && r->acc_steps[p] <= r->K;
      if (r->variant == MC_RULE_FLUX) {
        // np.round(cnt*((28-1)/(num_steps-1))).astype(int) != 11   (round-half-even)
        const double pos = std::nearbyint(r->cnt * ((28.0 - 1.0) / (r->num_steps - 1)));
        ok = ok && ((long)pos != 11);
      }
      if (r->variant == MC_RULE_FRAMEPACK) ok = ok && std::fabs(1.0 - cur) <= 0.06;   // :265
    }
    if (ok) {
      skip = true;
    } else {
      r->acc_err[p] = 0.0;
      r->acc_steps[p] = 0;
      r->acc_ratio[p] = 1.0;
    }
  }
  r->cnt += 1;
  if (r->cnt >= r->num_steps) {
    r->cnt = 0;
    // Qwen-Image and FramePack only rewind the counter (FramePack re-initialises at cnt == 0 instead)
    if (r->variant != MC_RULE_QWEN && r->variant != MC_RULE_FRAMEPACK)
    for (int i = 0; i < 2; ++i) {
      r->acc_ratio[i] = 1.0;
      r->acc_err[i] = 0.0;
      r->acc_steps[i] = 0;
    }
  }
  return skip ? 1 : 0;
}

int mc_rule_cnt(const mc_rule* r) { return r->cnt; }

void mc_rule_state(const mc_rule* r, double acc_err[2], int acc_steps[2], double acc_ratio[2]) {
  for (int i = 0; i < 2; ++i) {
    if (acc_err) acc_err[i] = r->acc_err[i];
    if (acc_steps) acc_steps[i] = r->acc_steps[i];
    if (acc_ratio) acc_ratio[i] = r->acc_ratio[i];
  }
}

// nearest_interp (MagCache4Wan2.1/magcache_generate.py:27-34): np.round is round-half-even
void mc_nearest_interp(const double* src, int src_len, double* dst, int target_length) {
  if (target_length == 1) {
    dst[0] = src[src_len - 1];
    return;
  }
  const double scale = (double)(src_len - 1) / (double)(target_length - 1);
  for (int i = 0; i < target_length; ++i) {
    long idx = (long)std::nearbyint((double)i * scale);
    dst[i] = src[idx];
  }
}

}  // extern "C"
