/* amt_post.h — C ABI of libamt_post.so: the CPU post-processing that follows the hot path.
 *
 * Native in the reference (Rust crate `modelutil`), therefore native (C++) here:
 *   stitch_probs            rust-plugins/src/common.rs:13-45    (python.rs:962-969)
 *   extract_events          rust-plugins/src/common.rs:47-144   (python.rs:971-978)
 *   convert_to_frame_events rust-plugins/src/python.rs:423-447  (python.rs:980-1005 to_frame_events)
 *   extract_midi_events / free_midi_events   rust-plugins/src/cbinds.rs:51-91 (same names, same
 *   #[repr(C)] structs: a caller of the reference's iOS library can link this one instead)
 * Integer / index results are bit-exact with the reference algorithm; stitched probabilities
 * follow its f64 blend arithmetic and f32 rounding.
 */
#ifndef AMT_POST_H_
#define AMT_POST_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cbinds.rs:9-22 */
typedef struct MidiEvent {
  uint64_t attack_time;
  uint8_t note;
  uint64_t duration;
  uint8_t velocity;
} MidiEvent;

typedef struct MidiEventList {
  MidiEvent* ptr;
  size_t length;
  size_t _capacity;
} MidiEventList;

/* cbinds.rs:24-29 (N = 3): strides and dims in ELEMENTS, data = IEEE half floats */
typedef struct MLMultiArrayWrapper3 {
  uint64_t strides[3];
  uint64_t dims[3];
  const uint8_t* data;
} MLMultiArrayWrapper3;

/* cbinds.rs:51-75: stitch the [windows][frames][events] f16 probabilities, extract events.
 * Caller frees with free_midi_events. */
MidiEventList* extract_midi_events(MLMultiArrayWrapper3 data, double overlap, double duration_per_frame);
void free_midi_events(MidiEventList* ptr);

/* frames produced by stitching (common.rs:20-22) */
int64_t amt_stitched_frames(int64_t num_windows, int64_t frames_per_window, double overlap,
                            double duration_per_frame);
/* all_probs [W][F][E] float32 contiguous -> out [amt_stitched_frames][E] float32.
 * Returns 0, or 1 if the reference would have indexed out of bounds (it panics there). */
int32_t amt_stitch_probs(const float* all_probs, int64_t num_windows, int64_t frames_per_window,
                         int64_t event_categories, double overlap, double duration_per_frame, float* out);
/* probs [frames][notes] float32 -> sorted events (attack_frame, key, duration, velocity) */
MidiEventList* amt_extract_events(const float* probs, int64_t num_frames, int64_t num_notes);
/* events as 4 x uint32 (attack_frame, key, frame_duration, velocity) -> frames [S][90] float32 */
int32_t amt_convert_to_frame_events(const uint32_t* events, int64_t num_events, int32_t model_output_size,
                                    int32_t start_frame, int32_t num_frames_with_backing_samples, float* out);

#ifdef __cplusplus
}
#endif
#endif /* AMT_POST_H_ */
