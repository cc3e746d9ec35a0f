// CPU post-processing behind include/amt_post.h: window stitching and note-event extraction.
// Follows rust-plugins/src/common.rs:13-144 and python.rs:423-447 statement by statement in
// behaviour (thresholds, look-ahead, f64 blend arithmetic, truncations), written for C++.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <tuple>
#include <vector>

#include "../../include/amt_post.h"

namespace {

constexpr int kNumEventTypes = 90;  // python.rs:47-51 NUM_EVENT_TYPES

float half_to_float(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else {  // subnormal
      int e = -1;
      do { ++e; man <<= 1; } while ((man & 0x400u) == 0);
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | (man << 13);
  } else {
    bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

// common.rs:13-45.  get(w, f, e) returns the probability as f32.
template <typename Get>
int stitch(Get get, int64_t W, int64_t F, int64_t E, double overlap, double dpf, float* out, int64_t out_frames) {
  const double overlapping_frames = overlap / dpf;
  std::fill(out, out + out_frames * E, 0.0f);
  double base = 0.0;
  const int64_t ceil_ov = (int64_t)std::ceil(overlapping_frames);
  for (int64_t w = 0; w < W; ++w) {
    for (int64_t f = 0; f < F; ++f) {
      const int64_t row = (int64_t)base + f;
      if (row < 0 || row >= out_frames) return 1;  // the reference panics on this index
      for (int64_t e = 0; e < E; ++e) {
        float* dst = out + row * E + e;
        const float p = get(w, f, e);
        if (w > 0 && f <= ceil_ov) {
          const double blend = (double)f / overlapping_frames;
          *dst = (float)((1.0 - blend) * (double)(*dst) + blend * (double)p);
        } else {
          *dst = p;
        }
      }
    }
    base += (double)F - overlapping_frames;
  }
  return 0;
}

int64_t stitched_frames(int64_t W, int64_t F, double overlap, double dpf) {
  const double overlapping_frames = overlap / dpf;
  return W * F - (int64_t)overlapping_frames * (W - 1);
}

typedef std::tuple<uint32_t, uint32_t, uint32_t, uint32_t> Ev;

// common.rs:47-144
std::vector<Ev> extract(const float* probs, int64_t num_frames, int64_t num_notes) {
  const float reactivation_gap = 0.1f, reactivation_threshold = 0.4f;
  const float activation_threshold = 0.5f, deactivation_threshold = 0.1f;
  std::vector<Ev> events;
  auto P = [&](int64_t f, int64_t k) { return probs[f * num_notes + k]; };
  auto duration = [](int64_t end_frame, int64_t start_frame) -> uint32_t {
    int32_t d = (int32_t)end_frame - (int32_t)start_frame;
    return (uint32_t)std::max(d, 1);
  };
  const uint32_t velocity = 7;  // common.rs:61-64 (TODO in the reference)
  struct Playing { bool on; int64_t started; float prob; };
  std::vector<Playing> playing(num_notes, Playing{false, 0, 0.f});
  for (int64_t frame = 0; frame < num_frames; ++frame) {
    for (int64_t key = 0; key < num_notes; ++key) {
      auto activation_prob = [&]() -> float {
        float a = P(frame, key);
        const int64_t lookahead = 10;
        for (int64_t i = frame + 1; i < num_frames; ++i) {
          if (P(i, key) > a) {
            a = P(i, key);
          } else if (i - frame > lookahead) {
            break;
          }
        }
        return a;
      };
      if (playing[key].on) {
        const int64_t started_at = playing[key].started;
        if (P(frame, key) < deactivation_threshold) {
          events.emplace_back((uint32_t)started_at, (uint32_t)key, duration(frame, started_at), velocity);
          playing[key].on = false;
        } else {
          const float since = (float)frame - (float)started_at;
          bool should_reactivate = false;
          if (since > 5.0f) {
            const int64_t samples = 6;
            float prev = 0.0f;
            for (int64_t i = frame - samples; i < frame; ++i) prev += P(i, key);
            prev /= (float)samples;
            float next = 0.0f;
            for (int64_t i = frame; i < std::min(frame + samples, num_frames); ++i) next += P(i, key);
            next /= (float)samples;
            should_reactivate = (next - prev) > reactivation_gap;
          }
          if (frame < num_frames - 1 && P(frame, key) < P(frame + 1, key)) continue;
          if (P(frame, key) > reactivation_threshold && should_reactivate) {
            events.emplace_back((uint32_t)started_at, (uint32_t)key, duration(frame - 1, started_at), velocity);
            playing[key] = Playing{true, frame, activation_prob()};
          }
        }
      } else if (P(frame, key) > activation_threshold) {
        playing[key] = Playing{true, frame, activation_prob()};
      }
    }
  }
  for (int64_t key = 0; key < num_notes; ++key) {
    if (playing[key].on) {
      events.emplace_back((uint32_t)playing[key].started, (uint32_t)key, duration(num_frames, playing[key].started),
                          velocity);
      playing[key].on = false;
    }
  }
  std::stable_sort(events.begin(), events.end());
  return events;
}

MidiEventList* to_list(const std::vector<Ev>& ev) {
  MidiEventList* l = (MidiEventList*)malloc(sizeof(MidiEventList));
  l->length = ev.size();
  l->_capacity = ev.size();
  l->ptr = ev.empty() ? nullptr : (MidiEvent*)malloc(sizeof(MidiEvent) * ev.size());
  for (size_t i = 0; i < ev.size(); ++i) {
    memset(&l->ptr[i], 0, sizeof(MidiEvent));
    l->ptr[i].attack_time = std::get<0>(ev[i]);
    l->ptr[i].note = (uint8_t)std::get<1>(ev[i]);
    l->ptr[i].duration = std::get<2>(ev[i]);
    l->ptr[i].velocity = (uint8_t)std::get<3>(ev[i]);
  }
  return l;
}

}  // namespace

extern "C" {

int64_t amt_stitched_frames(int64_t W, int64_t F, double overlap, double dpf) { return stitched_frames(W, F, overlap, dpf); }

int32_t amt_stitch_probs(const float* all_probs, int64_t W, int64_t F, int64_t E, double overlap, double dpf,
                         float* out) {
  const int64_t frames = stitched_frames(W, F, overlap, dpf);
  if (frames < 0) return 1;
  return stitch([&](int64_t w, int64_t f, int64_t e) { return all_probs[(w * F + f) * E + e]; }, W, F, E, overlap, dpf,
                out, frames);
}

MidiEventList* amt_extract_events(const float* probs, int64_t num_frames, int64_t num_notes) {
  return to_list(extract(probs, num_frames, num_notes));
}

MidiEventList* extract_midi_events(MLMultiArrayWrapper3 data, double overlap, double duration_per_frame) {
  const int64_t W = (int64_t)data.dims[0], F = (int64_t)data.dims[1], E = (int64_t)data.dims[2];
  const uint16_t* h = reinterpret_cast<const uint16_t*>(data.data);
  const int64_t frames = stitched_frames(W, F, overlap, duration_per_frame);
  std::vector<float> stitched((size_t)std::max<int64_t>(frames, 0) * E);
  int rc = stitch([&](int64_t w, int64_t f, int64_t e) {
    return half_to_float(h[w * (int64_t)data.strides[0] + f * (int64_t)data.strides[1] + e * (int64_t)data.strides[2]]);
  }, W, F, E, overlap, duration_per_frame, stitched.data(), frames);
  if (rc != 0) abort();  // the reference panics (aborts across the FFI) on this condition
  return to_list(extract(stitched.data(), frames, E));
}

void free_midi_events(MidiEventList* ptr) {
  if (!ptr) return;
  free(ptr->ptr);
  free(ptr);
}

int32_t amt_convert_to_frame_events(const uint32_t* events, int64_t num_events, int32_t model_output_size,
                                    int32_t start_frame, int32_t num_frames_with_backing_samples, float* out) {
  // python.rs:423-447
  std::fill(out, out + (size_t)model_output_size * kNumEventTypes, 0.0f);
  for (int64_t i = 0; i < num_events; ++i) {
    const uint32_t attack = events[4 * i], key = events[4 * i + 1], dur = events[4 * i + 2];
    if (key >= (uint32_t)kNumEventTypes) return 1;
    const int32_t frame_start = (int32_t)attack - start_frame;
    const int32_t frame_end = frame_start + (int32_t)dur;
    if (frame_start > 0 && frame_start < model_output_size) out[(size_t)(frame_start - 1) * kNumEventTypes + key] = 0.0f;
    const int32_t lo = std::max(frame_start, 0);
    const int32_t hi = std::min(std::min(frame_end, model_output_size), num_frames_with_backing_samples);
    for (int32_t f = lo; f < hi; ++f) {
      const float t = (float)f - (float)frame_start;
      out[(size_t)f * kNumEventTypes + key] = std::max(std::exp(-0.05f * t), 0.6f);
    }
  }
  return 0;
}

}  // extern "C"
