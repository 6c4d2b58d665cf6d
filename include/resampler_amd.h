/*
 * resampler_amd.h -- C ABI of the MI355X-native resampling engine (libresampler_amd.so).
 *
 * Drop-in boundary for the hot path of hasenbanck/resampler v0.5.1: each entry point names the
 * reference interface it replaces (file:line relative to the reference repository).  Plain
 * pointers and sizes only; all buffers are interleaved f32 frames, all counts are numbers of f32
 * values across all channels exactly as in the reference (src/resampler_fir.rs:617-620).
 *
 * Pointers named `in`/`out` are HOST pointers; pointers named `d_in`/`d_out` are DEVICE (HBM)
 * pointers on the handle's device.  `stream` is a hipStream_t passed as void*.  NULL = the handle's
 * (or batch's) OWN stream, created hipStreamNonBlocking: work enqueued there is NOT ordered against
 * the legacy default stream -- a caller whose buffers are produced / consumed on the default stream
 * passes RSMP_STREAM_LEGACY (== hipStreamLegacy) instead, or a stream of its own and orders that.
 * Device entry points are asynchronous on `stream`; the returned counts
 * are exact and available immediately (they come from the host-side mirror of the reference
 * state machine, not from the GPU).
 *
 * There is no CPU fallback: every compute entry point fails with RSMP_ERR_NO_DEVICE when no HIP
 * device is present.  The rsmp_*_plan_* / rsmp_design_* entry points are host-only (filter
 * design, FFT planning, and the (consumed, produced) state machine) and work without a GPU.
 */
#ifndef RESAMPLER_AMD_H
#define RESAMPLER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The legacy default stream as a `stream` argument (the value of hipStreamLegacy); NULL does not mean it. */
#define RSMP_STREAM_LEGACY ((void*)1)

/* ---- status codes: 1 and 2 are ResampleError (src/error.rs:3-8) ------------------------------- */
enum {
    RSMP_OK = 0,
    RSMP_ERR_INVALID_INPUT_BUFFER_SIZE = 1,  /* ResampleError::InvalidInputBufferSize  */
    RSMP_ERR_INVALID_OUTPUT_BUFFER_SIZE = 2, /* ResampleError::InvalidOutputBufferSize */
    RSMP_ERR_INVALID_ARGUMENT = 3,           /* where the reference panics (resampler_fir.rs:302-309) */
    RSMP_ERR_NO_DEVICE = 4,
    RSMP_ERR_HIP = 5,
    RSMP_ERR_CAPACITY = 6                    /* caller-provided array too small (bulk/plan calls) */
};

/* enum SampleRate (src/lib.rs:167-188), same order as the reference. */
enum {
    RSMP_HZ22050 = 0, RSMP_HZ16000, RSMP_HZ32000, RSMP_HZ44100, RSMP_HZ48000,
    RSMP_HZ88200, RSMP_HZ96000, RSMP_HZ176400, RSMP_HZ192000, RSMP_HZ384000
};
/* enum Latency (src/resampler_fir.rs:139-149): taps = 16, 32, 64, 128. */
enum { RSMP_LATENCY_SAMPLE8 = 0, RSMP_LATENCY_SAMPLE16, RSMP_LATENCY_SAMPLE32, RSMP_LATENCY_SAMPLE64 };
/* enum Attenuation (src/resampler_fir.rs:102-110): Kaiser beta = 7, 10, 13. */
enum { RSMP_ATTENUATION_DB60 = 0, RSMP_ATTENUATION_DB90, RSMP_ATTENUATION_DB120 };

/* FIR kernel selection (rsmp_fir_set_kernel).  AUTO picks PERIODIC when the rate pair reduces to
 * a small rational and the launch is long enough, GENERIC otherwise.  PERIODIC runs streams
 * on the matrix cores where the geometry allows: the split kernel (every f32 operand as the sum
 * of two fp16 values of the scaled operand, three fp16 MFMA products accumulated in f32; RSMP_FIR_SPLIT_PLANES=3
 * selects three bf16 planes / six products, exact in every bit) for rate pairs with
 * 16..160 classes such as 44.1 <-> 48 kHz and 1 .. 16 channels (channel pairs), the exact-f32 MFMA
 * kernel for other 2-channel geometries, the vector kernel otherwise.  PERIODIC_F32 keeps
 * every product in f32 (exact-f32 MFMA or vector kernels, never the split one); PERIODIC_VECTOR forces
 * the packed-FMA vector kernel for every channel count.  All produce the same results within the
 * 1e-6 RMS gate (measured against the CPU path: about 1.2e-7 RMS each).
 * Non-finite and out-of-range input: every kernel gives the reference's answer -- the same outputs are
 * finite, +inf, -inf or NaN (src/fir/avx.rs:25-58).  The PERIODIC kernels check the sums they store; a
 * launch that saw a non-finite sum (an inf / NaN sample, or in the split kernel a sample of magnitude
 * >= 16, beyond its fp16 planes) is followed by a repair launch that re-evaluates the affected 1024-frame
 * chunks in the reference's own two-row form (tests/test_fir_gpu.py::
 * test_non_finite_and_huge_input_match_the_reference).  One residue: at outputs whose exact position is a
 * multiple of 1/1024 frame the reference's inf-versus-NaN depends on the sign of an ~1e-12 rounding residue
 * of its f64 position; the repair pass may say NaN where the reference says inf or vice versa (GENERIC, which
 * replays the exact f64 positions, does not).  Clean audio never takes the repair path. */
enum { RSMP_FIR_KERNEL_AUTO = 0, RSMP_FIR_KERNEL_GENERIC = 1, RSMP_FIR_KERNEL_PERIODIC = 2,
       RSMP_FIR_KERNEL_PERIODIC_VECTOR = 3, RSMP_FIR_KERNEL_PERIODIC_F32 = 4 };

const char* rsmp_last_error(void);          /* thread-local message of the last failing call */
int rsmp_device_count(void);                /* number of HIP devices, 0 when there is none   */
const char* rsmp_version(void);
/* impl From<SampleRate> for u32 (src/lib.rs:219-236); 0 for an invalid enum value. */
uint32_t rsmp_sample_rate_hz(int sample_rate);

/* ============================ ResamplerFir (src/resampler_fir.rs) =============================== */
typedef struct rsmp_fir rsmp_fir;

/* ResamplerFir::new (resampler_fir.rs:252-266).  NULL + rsmp_last_error() on failure. */
rsmp_fir* rsmp_fir_new(size_t channels, int input_rate, int output_rate, int latency,
                       int attenuation, int device);
/* ResamplerFir::new_from_hz (resampler_fir.rs:295-404); zero rates -> NULL (reference panics). */
rsmp_fir* rsmp_fir_new_from_hz(size_t channels, uint32_t input_rate_hz, uint32_t output_rate_hz,
                               int latency, int attenuation, int device);
/* Drop (resampler_fir.rs:61-67 et al.). */
void rsmp_fir_free(rsmp_fir* r);
/* ResamplerFir::buffer_size_output (resampler_fir.rs:456-465). */
size_t rsmp_fir_buffer_size_output(const rsmp_fir* r);
/* ResamplerFir::delay (resampler_fir.rs:630-632). */
size_t rsmp_fir_delay(const rsmp_fir* r);
/* ResamplerFir::reset (resampler_fir.rs:638-642). */
void rsmp_fir_reset(rsmp_fir* r);
/* fmt::Debug fields (resampler_fir.rs:203-211). */
size_t rsmp_fir_channels(const rsmp_fir* r);
size_t rsmp_fir_taps(const rsmp_fir* r);
size_t rsmp_fir_phases(const rsmp_fir* r);
/* The reference's private streaming state (resampler_fir.rs:189-192: read_position, available_frames,
 * position), for tests and checkpoints: it must equal the reference's after the same calls. */
void rsmp_fir_state(const rsmp_fir* r, size_t* read_position, size_t* available_frames, double* position);
int rsmp_fir_set_kernel(rsmp_fir* r, int kernel);
/* Measurement hook: when enabled, the convolution launch(es) of every call made through this
 * handle (the first handle of a batch call) are bracketed by HIP events on the launch stream;
 * rsmp_fir_last_kernel_ms waits for the last one and returns its duration. */
int rsmp_fir_set_profiling(rsmp_fir* r, int enable);
int rsmp_fir_last_kernel_ms(rsmp_fir* r, float* ms);
/* Mean over the (up to 64) most recent launches made since profiling was enabled; no host sync
 * happens between those launches, so this is the kernel's duration inside a timed region. */
int rsmp_fir_mean_kernel_ms(rsmp_fir* r, float* ms, size_t* launches);
/* Which kernel the handle's last launch used (diagnostic, for benchmark reports): 0 generic
 * (any ratio), 1 periodic vector kernel, 2 periodic vector kernel with double-buffered
 * workgroups, 3 periodic exact-f32 matrix-core kernel, 4 periodic split matrix-core kernel with
 * three bf16 planes per operand (RSMP_FIR_SPLIT_PLANES=3), 5 periodic split matrix-core kernel with
 * two fp16 planes per operand (the default for rate pairs it has a geometry for); negative: invalid
 * handle. */
int rsmp_fir_kernel_variant(const rsmp_fir* r);

/* ResamplerFir::resample (resampler_fir.rs:509-621): one call, host buffers, synchronous. */
int rsmp_fir_resample(rsmp_fir* r, const float* in, size_t in_len, float* out, size_t out_len,
                      size_t* consumed, size_t* produced);
/* Same call with HBM-resident buffers, asynchronous on `stream`. */
int rsmp_fir_resample_device(rsmp_fir* r, const float* d_in, size_t in_len, float* d_out,
                             size_t out_len, size_t* consumed, size_t* produced, void* stream);

/* Bulk form: the result is DEFINED as what the reference driver loop resample_batch_fir
 * (resample/src/main.rs:226-254) returns when it feeds `chunk_len`-value slices (the CLI uses
 * 512) through resample() with a buffer_size_output() scratch: same output values, same total
 * count, same per-call (consumed, produced) pairs (written to calls[2*i], calls[2*i+1] for the
 * first max_calls calls; calls may be NULL).  One kernel launch for the whole buffer.
 * out_cap must hold every produced value (RSMP_ERR_CAPACITY otherwise; use
 * rsmp_fir_bulk_output_bound). */
size_t rsmp_fir_bulk_output_bound(const rsmp_fir* r, size_t in_len, size_t chunk_len);
int rsmp_fir_resample_bulk(rsmp_fir* r, const float* in, size_t in_len, size_t chunk_len,
                           float* out, size_t out_cap, size_t* consumed, size_t* produced,
                           size_t* calls, size_t max_calls, size_t* n_calls);
int rsmp_fir_resample_bulk_device(rsmp_fir* r, const float* d_in, size_t in_len, size_t chunk_len,
                                  float* d_out, size_t out_cap, size_t* consumed, size_t* produced,
                                  size_t* calls, size_t max_calls, size_t* n_calls, void* stream);

/* Batch form: n independent resampler instances (all on the same device), stream i processing
 * d_in[i] -> d_out[i] with bulk semantics, all in ONE launch per kernel flavour.  This is the
 * unit that is sharded across GPUs (one process per GPU, a contiguous range of streams each). */
int rsmp_fir_batch_resample_bulk_device(rsmp_fir* const* rs, size_t n, const float* const* d_in,
                                        const size_t* in_lens, size_t chunk_len,
                                        float* const* d_out, const size_t* out_caps,
                                        size_t* consumed, size_t* produced, void* stream);
/* The same with the PLANNER named.  A launch plans every distinct state among its streams on the host (streams in one state that are
 * fed the same amount share a plan): ~30 us of a core per state and thousand calls -- 2-3 ms for 64 streams x 4096 calls in 64 states
 * around a 0.3 ms kernel.  planner = 1 plans on the DEVICE instead (rsmp_fir_lockstep_run_bulk on a lock-step batch over the same
 * handles, which the library keeps from launch to launch for as long as the handles are touched through nothing else; the states are
 * written back into the handles before the call returns, so the call returns when the launch is THROUGH, not when it is enqueued);
 * planner = 0 never; planner = -1 -- what rsmp_fir_batch_resample_bulk_device does -- where the batch has at least 16 different
 * states.  The device planner takes batches of one channel count and one buffer length, calls of at most 2048 frames, at least eight
 * of them, and output buffers of rsmp_fir_bulk_output_bound; anything else is planned on the host whatever `planner` says.
 * *planned_on_device (may be null): which it was.  Same (consumed, produced), samples and end states either way. */
int rsmp_fir_batch_resample_bulk_device_ex(rsmp_fir* const* rs, size_t n, const float* const* d_in,
                                           const size_t* in_lens, size_t chunk_len,
                                           float* const* d_out, const size_t* out_caps,
                                           size_t* consumed, size_t* produced, void* stream, int planner, int* planned_on_device);
/* The same over a two-channel WAV file's samples as they are in the file (resample/src/main.rs:128-137): d_pcm[i] =
 * little-endian PCM of `bits` (16 / 24 / 32) per sample, in_lens[i] SAMPLES (2 per frame), 4-byte aligned.  The samples are
 * converted where the kernels read their input (`sample as f32 / (1 << (bits - 1)) as f32`, with the reference's 32-bit
 * divisor -2^31): the output equals rsmp_pcm_to_stereo_f32_device + rsmp_fir_batch_resample_bulk_device's within the
 * kernels' usual tolerance (the same samples reach the same arithmetic), and the launch reads 2 - 4 bytes a sample
 * instead of the conversion pass's PCM + 4 written + 4 read.  Long launches need the 128-tap rate pairs of the split
 * kernel (44.1 <-> 48 kHz, 96 -> 44.1 / 48 kHz ...: RSMP_ERR_INVALID_ARGUMENT otherwise -- convert first); at most one
 * launch's worth of input per stream (46 M outputs). */
int rsmp_fir_batch_resample_bulk_pcm_device(rsmp_fir* const* rs, size_t n, const void* const* d_pcm, int bits,
                                            const size_t* in_lens, size_t chunk_len, float* const* d_out,
                                            const size_t* out_caps, size_t* consumed, size_t* produced, void* stream);

/* reset() for every stream of a batch. */
void rsmp_fir_batch_reset(rsmp_fir* const* rs, size_t n);

/* ---- lock-step batch: a fixed set of streams, ONE resample() call per stream and step ------------
 * BASELINE config 4 (1024 mixed-rate streams, one 512-frame chunk each per step).  Per stream a step
 * is exactly ResamplerFir::resample (resampler_fir.rs:509-621) on d_in[i] + in_offset_frames*channels_i
 * (in_frames frames) into d_out[i] (or, with `append`, behind what the earlier steps wrote there): same
 * frames accepted, same output, same frames kept.  Unlike the calls above the reference's state
 * (read_position / available_frames / position, :189-192) lives in HBM and the control flow runs inside
 * the kernel, one lane per stream: a step is a single launch with constant arguments and no per-stream
 * host work, whatever states the streams are in.  Consequently the (consumed, produced) counts of a step
 * are device data: rsmp_fir_lockstep_counts waits for the last step and copies them out.
 * While a batch exists its streams must not be used through other entry points; rsmp_fir_lockstep_sync
 * (and _free) write the device state back into the handles.  out_caps[i] must be at least
 * rsmp_fir_buffer_size_output (the reference's documented sizing).  Steps of one batch are ordered. */
typedef struct rsmp_fir_lockstep rsmp_fir_lockstep;
rsmp_fir_lockstep* rsmp_fir_lockstep_new(rsmp_fir* const* rs, size_t n, size_t max_step_frames);
void rsmp_fir_lockstep_free(rsmp_fir_lockstep* ls);
size_t rsmp_fir_lockstep_size(const rsmp_fir_lockstep* ls);
size_t rsmp_fir_lockstep_workgroups(const rsmp_fir_lockstep* ls);   /* diagnostic: workgroups per step */
/* ... and how many of them use split operands on the fp16 matrix cores (two-channel streams in
 * RSMP_FIR_KERNEL_AUTO; the others keep every product in f32: other channel counts, streams set to another
 * kernel mode with rsmp_fir_set_kernel before the batch was made, geometries whose image does not fit). */
size_t rsmp_fir_lockstep_split_workgroups(const rsmp_fir_lockstep* ls);
/* Binding (again) starts the `append` positions at the front of the new output buffers.  With `append` the
 * caller sizes d_out[i] for all the steps it will run until the next bind (the kernel checks the room of one
 * step, out_caps[i], not of the buffer). */
int rsmp_fir_lockstep_bind(rsmp_fir_lockstep* ls, const float* const* d_in, float* const* d_out,
                           const size_t* out_caps);
/* d_in_frames: optional DEVICE array of frames offered per stream (in the order of `rs`; NULL = in_frames
 * for every stream; entries above max_step_frames are clamped). */
int rsmp_fir_lockstep_step(rsmp_fir_lockstep* ls, size_t in_frames, size_t in_offset_frames,
                           const uint32_t* d_in_frames, int append, void* stream);
int rsmp_fir_lockstep_counts(rsmp_fir_lockstep* ls, size_t* consumed, size_t* produced);
/* k_steps consecutive steps in one go: per stream the reference's driver loop (resample/src/main.rs:226-254) over the
 * k_steps * in_frames frames at d_in[i] + in_offset_frames, one resample() call (src/resampler_fir.rs:509-621) per
 * in_frames frames -- the same calls, counts, samples and end state as k_steps calls of rsmp_fir_lockstep_step with
 * `append` at offsets in_offset_frames + s * in_frames.  The calls are planned on the device (one lane per stream
 * replays the reference's control flow for the k_steps calls) and computed by the bulk kernels, one launch per rate
 * pair for the whole run, instead of k_steps launches of the one-call kernel.  The outputs of the calls follow each
 * other in d_out[i]: behind what was appended before (append != 0), or from the front of the buffer (append == 0: the
 * append position starts again there); the caller sizes d_out[i] for the run.  rsmp_fir_lockstep_counts returns the
 * last call's counts, rsmp_fir_lockstep_run_counts all of them: consumed[s * n + i], produced[s * n + i] for call s
 * of stream i (up to max_steps calls).  Batches with a rate pair no bulk kernel serves (irrational ratios) and runs
 * of one call are executed as a loop of steps. */
int rsmp_fir_lockstep_run(rsmp_fir_lockstep* ls, size_t k_steps, size_t in_frames, size_t in_offset_frames,
                          int append, void* stream);
int rsmp_fir_lockstep_run_counts(rsmp_fir_lockstep* ls, size_t* consumed, size_t* produced, size_t max_steps);
/* A whole buffer per stream as the reference's driver loop feeds it (resample/src/main.rs:226-254): total_frames frames
 * from d_in[i] + in_offset_frames in calls of chunk_frames frames, the last call shorter where total_frames is no
 * multiple -- floor(total / chunk) calls through rsmp_fir_lockstep_run and the remaining frames as one more call, its
 * output appended behind theirs.  Asynchronous on `stream`, planned on the device whatever states the streams are in
 * (no host planning, no host threads): the bulk entry point for batches of streams in DISTINCT states.  Counts:
 * rsmp_fir_lockstep_run_counts (the equal calls) and rsmp_fir_lockstep_counts (the last call); rsmp_fir_lockstep_sync
 * brings the streams' states back into their handles.  chunk_frames <= the batch's max_step_frames, and calls every stream
 * accepts WHOLE: chunk_frames + taps + 8 <= 4096 (INPUT_CAPACITY, resampler_fir.rs:18; the driver loop offers a call's
 * remainder again, a run's calls read at fixed offsets) -- RSMP_ERR_INVALID_INPUT_BUFFER_SIZE otherwise. */
int rsmp_fir_lockstep_run_bulk(rsmp_fir_lockstep* ls, size_t total_frames, size_t chunk_frames, size_t in_offset_frames,
                               int append, void* stream);
/* Diagnostic: calls of the last run (all streams) that the device planner's fast path declined and the plain state
 * machine did (fir_mirror_fast.h); 0 for a run executed as a loop of steps. */
int rsmp_fir_lockstep_run_slow_calls(rsmp_fir_lockstep* ls, size_t* slow_calls);
/* Diagnostic: how often a class of the batch's streams was given new class tables because the streams' f64 position drift
 * (src/resampler_fir.rs:589: ~1e-14 of a frame per output, for as long as a stream runs) had moved away from the drift
 * its tables were mixed for.  The batch watches the drifts itself (an asynchronous read-back every 2^19 frames per
 * stream); nothing for the caller to do. */
int rsmp_fir_lockstep_table_rebinds(const rsmp_fir_lockstep* ls, size_t* rebinds);
/* Diagnostic counters of a batch, out[0 .. n): [0] table rebinds (as above), [1] runs taken over from the plan stream
 * (planned while the run before them computed), [2] runs planned ahead and dropped (the caller did something else),
 * [3] looks at the drifts that found a class past its tolerance with its next tables still on their way (the old ones
 * serve until the next look; a replacement is prepared by a worker thread, never inside a launch call), [4] times a
 * launch call WAITED for that worker (a class three tolerances past its tables: never observed), [5] probes of which
 * plan stream runs beside a caller's stream, [6] 1 if the last run's stream has a plan stream, [7] drift classes. */
#define RSMP_LS_STAT_COUNT 8
int rsmp_fir_lockstep_stats(const rsmp_fir_lockstep* ls, uint64_t* out, size_t n);
/* How closely the class tables follow the streams' drift: a class gets new tables when its drift is more than
 * `tolerance_frames` (default 1.2e-7: at worst 2e-7 of a full-scale sample, a fifth of the 1e-6 bound; 2e-8 .. 1e-6) from
 * what its tables were mixed for; the drifts are read back every `check_frames` input frames per stream (default 2^19).
 * Tighter = closer to the reference's phase rows, more replacements (each is prepared off the launch path). */
int rsmp_fir_lockstep_set_drift_policy(rsmp_fir_lockstep* ls, double tolerance_frames, size_t check_frames);
/* Sticky per-stream flags: 1 = more position runs in one step than the kernel keeps (outputs of that step
 * undefined; never observed), 2 = a step saw non-finite samples and was evaluated in the reference's
 * two-row form, 4 = the f64 position drifted out of the class tables' tolerance (reference form from then on),
 * 8 = a call of a run accepted fewer frames than it was offered (cannot happen below 3960 frames per step),
 * 16 = rsmp_fir_lockstep_run's planner found, replaying a call, a premise of its closed form violated (a drift of the f64
 * position beyond 1 / out_hz; never observed: the run's counts are then not the reference's). */
int rsmp_fir_lockstep_status(rsmp_fir_lockstep* ls, uint32_t* status);
int rsmp_fir_lockstep_sync(rsmp_fir_lockstep* ls);
/* rsmp_fir_lockstep_sync + what every stream has done since the batch's states were last exchanged with the handles (creation, reset,
 * the previous sync): accepted[i] = frames accepted x channels, produced[i] = frames produced x channels -- the (consumed, produced) a
 * bulk driver loop over the same input would return (resample/src/main.rs:226-254: every call accepts its whole offer) --, *status_or =
 * the OR of the streams' sticky flags.  rsmp_fir_lockstep_in_sync: *in_sync = 1 if no handle has been touched through another entry
 * point since (its state and history buffer are what the batch last wrote back): a host that alternates between the batch and other
 * entries re-creates the batch when this says 0.  rsmp_fir_batch_distinct_states: how many different states n handles are in (what a
 * host-planned bulk launch has to plan: streams in one state and fed the same amount share a plan). */
/* New buffers under a bound batch -- the same capacities, states and histories; what rsmp_fir_lockstep_bind does when only d_in / d_out
 * change, without its waits: the table goes to the device in `stream` order, and a run planned ahead (rsmp_fir_lockstep_run: the next
 * run of a caller that feeds run after run) survives if it starts at the front of the output -- the two pointers of its descriptors are
 * patched when it is taken over.  A service that hands every launch fresh buffers keeps the planner off its critical path that way. */
int rsmp_fir_lockstep_rebind_buffers(rsmp_fir_lockstep* ls, const float* const* d_in, float* const* d_out, void* stream);
/* rsmp_fir_lockstep_free without the write-back: for a batch whose handles have been used through other entry points since its last
 * sync (the handles hold the newer state). */
void rsmp_fir_lockstep_discard(rsmp_fir_lockstep* ls);
int rsmp_fir_lockstep_sync_totals(rsmp_fir_lockstep* ls, size_t* accepted, size_t* produced, uint32_t* status_or);
int rsmp_fir_lockstep_in_sync(const rsmp_fir_lockstep* ls, int* in_sync);
int rsmp_fir_batch_distinct_states(rsmp_fir* const* rs, size_t n, size_t* distinct);
/* Measurement hooks, as rsmp_fir_set_profiling / rsmp_fir_mean_kernel_ms: HIP events on the launch stream
 * around every step while enabled; the mean covers the (up to 64) most recent steps. */
int rsmp_fir_lockstep_set_profiling(rsmp_fir_lockstep* ls, int enable);
int rsmp_fir_lockstep_mean_kernel_ms(rsmp_fir_lockstep* ls, float* ms, size_t* launches);
/* ... and each of the last min(cap, 256) profiled launches' device time, oldest first (median / max of a bench) */
int rsmp_fir_lockstep_kernel_ms(rsmp_fir_lockstep* ls, float* ms, size_t cap, size_t* launches);
int rsmp_fir_lockstep_reset(rsmp_fir_lockstep* ls);   /* reset() of every stream (resampler_fir.rs:638-642) */

/* ---- host-only: filter design and the (consumed, produced) state machine ----------------------- */
/* make_sincs_for_kaiser table exactly as ResamplerFir::create_fir_coeffs lays it out
 * (resampler_fir.rs:406-422, window.rs:17-55): out[1024][taps]. */
int rsmp_design_fir_coeffs(uint32_t input_rate_hz, uint32_t output_rate_hz, int latency,
                           int attenuation, float* out, size_t out_len);
/* calculate_cutoff_kaiser (window.rs:114-131). */
double rsmp_design_cutoff_kaiser(size_t sample_count, double beta);

typedef struct rsmp_fir_plan rsmp_fir_plan;   /* host mirror of {read_position, available_frames, position} */
typedef struct {
    uint32_t out_start;   /* index of the first output frame of the run inside the launch      */
    uint32_t count;       /* output frames in the run                                          */
    int64_t in_base;      /* input frame (relative to the first buffered frame at launch start) */
                          /* that local position 0.0 refers to                                  */
    double p0;            /* position of the run's first frame (resampler_fir.rs:544)          */
    double inc;           /* exact position increment inside the run: p_k = p0 + k*inc         */
} rsmp_fir_segment;

rsmp_fir_plan* rsmp_fir_plan_new(uint32_t input_rate_hz, uint32_t output_rate_hz, int latency);
void rsmp_fir_plan_free(rsmp_fir_plan* p);
void rsmp_fir_plan_reset(rsmp_fir_plan* p);
void rsmp_fir_plan_state(const rsmp_fir_plan* p, size_t* read_position, size_t* available_frames,
                         double* position);
/* One reference resample() call in frames: how many input frames are accepted, how many output
 * frames are produced, and the exact position runs (segs may be NULL). */
int rsmp_fir_plan_call(rsmp_fir_plan* p, size_t input_frames, size_t output_capacity_frames,
                       size_t* frames_accepted, size_t* frames_produced, rsmp_fir_segment* segs,
                       size_t max_segs, size_t* n_segs);

/* A copy of the plan (its state); NULL for NULL. */
rsmp_fir_plan* rsmp_fir_plan_clone(const rsmp_fir_plan* p);
/* The driver loop of resample/src/main.rs:226-254 on the plan alone: calls of min(chunk_frames, remaining)
 * input frames with the full output capacity until in_frames are used up or max_calls calls were made
 * (0 = no limit).  Totals in frames.  What rsmp_fir_resample_bulk would do to a stream in this state. */
int rsmp_fir_plan_bulk(rsmp_fir_plan* p, size_t in_frames, size_t chunk_frames, size_t max_calls,
                       size_t* frames_accepted, size_t* frames_produced, size_t* n_calls);
/* Diagnostic: the device planner's arithmetic for runs of equal calls (rsmp_fir_lockstep_run; fir_mirror_fast.h:
 * structure predicted in exact integer arithmetic, the f64 chain verified against it) run on the host next to the
 * plain state machine for `calls` calls of in_frames frames, predicted in runs of run_len calls; compares counts,
 * every bit of the state and the outputs at integer positions call by call.  *mismatches must come back 0;
 * *slow_calls = calls the fast path declined (done by the plain state machine), *lean_calls = calls that were also done
 * by the unchecked chain the device takes where the prediction has no tie.  The plan advances by the calls. */
int rsmp_fir_plan_selftest_fast(rsmp_fir_plan* p, size_t in_frames, size_t calls, size_t run_len,
                                size_t* mismatches, size_t* slow_calls, size_t* lean_calls);
/* Starts a resampler in the middle of a stream: the handle takes the plan's state (same rate pair and
 * latency) and, as its buffered frames, the last available_frames * channels values of `history` -- the
 * input that precedes the point (host or device memory).  A stream cut at call boundaries found with
 * rsmp_fir_plan_bulk can so be resampled piece by piece, on different GPUs, with the output of one pass
 * (the reference has no counterpart: its state is private, resampler_fir.rs:189-192). */
int rsmp_fir_seek(rsmp_fir* r, const rsmp_fir_plan* p, const float* history, size_t history_len,
                  int history_on_device, void* stream);

/* ============================ CLI helpers around the path (resample/src) ========================== */
/* InterpolationResampler (resample/src/interpolation_resampler.rs:41-126): the comparison interpolators of
 * the reference's command-line tool, whole buffer in, ceil(frames * out / in) frames out. */
enum { RSMP_INTERP_LINEAR = 0, RSMP_INTERP_HERMITE = 1 };
size_t rsmp_interp_output_len(size_t channels, uint32_t in_hz, uint32_t out_hz, size_t in_len);   /* in f32 values */
int rsmp_interp_resample(int mode, size_t channels, uint32_t in_hz, uint32_t out_hz, const float* in,
                         size_t in_len, float* out, size_t out_cap, size_t* produced);
int rsmp_interp_resample_device(int mode, size_t channels, uint32_t in_hz, uint32_t out_hz, const float* d_in,
                                size_t in_len, float* d_out, size_t out_cap, size_t* produced, void* stream);
/* WAV sample conversion of the tool (resample/src/main.rs:128-156): little-endian integer PCM (16, packed
 * 24, 32 bits) -> f32 = s / 2^(bits-1), mono duplicated to both channels; d_out holds n_samples values
 * (2 channels in) or 2 * n_samples (1 channel in).  Device pointers, asynchronous on `stream`. */
int rsmp_pcm_to_stereo_f32_device(const void* d_pcm, int bits, int channels, size_t n_samples, float* d_out,
                                  void* stream);
/* Measurement aid (no counterpart in the reference): a plain streaming copy of n_values floats, 16 bytes per lane --
 * the rate a kernel that reads as much as it writes can reach on this GPU; bench.py reports it next to every
 * roofline fraction.  16-byte aligned device pointers, n_values a multiple of 4, asynchronous on `stream`. */
int rsmp_device_stream_copy(const float* d_src, float* d_dst, size_t n_values, void* stream);

/* ============================ ResamplerFft (src/resampler_fft.rs) =============================== */
typedef struct rsmp_fft rsmp_fft;

/* ResamplerFft::new (resampler_fft.rs:75-119); rates are SampleRate enum values.  NULL on
 * failure.  Channels are processed independently ("reference quirks"). */
rsmp_fft* rsmp_fft_new(size_t channels, int input_rate, int output_rate, int device);
void rsmp_fft_free(rsmp_fft* r);
/* ResamplerFft::chunk_size_input / chunk_size_output (resampler_fft.rs:135-145). */
size_t rsmp_fft_chunk_size_input(const rsmp_fft* r);
size_t rsmp_fft_chunk_size_output(const rsmp_fft* r);
/* ResamplerFft::delay (resampler_fft.rs:151-153). */
size_t rsmp_fft_delay(const rsmp_fft* r);
size_t rsmp_fft_channels(const rsmp_fft* r);
int rsmp_fft_set_profiling(rsmp_fft* r, int enable);
int rsmp_fft_last_kernel_ms(rsmp_fft* r, float* ms);

/* ResamplerFft::resample (resampler_fft.rs:182-240): one chunk; in_len >= chunk_size_input and
 * out_len >= chunk_size_output, extra values ignored (:186-192).  Host buffers, synchronous. */
int rsmp_fft_resample(rsmp_fft* r, const float* in, size_t in_len, float* out, size_t out_len);
int rsmp_fft_resample_device(rsmp_fft* r, const float* d_in, size_t in_len, float* d_out,
                             size_t out_len, void* stream);
/* n_chunks back-to-back chunks == n_chunks consecutive resample() calls (the whole-chunk loop of
 * resample_batch, resample/src/main.rs:276-289), one launch. */
int rsmp_fft_resample_bulk(rsmp_fft* r, const float* in, size_t in_len, float* out, size_t out_len,
                           size_t n_chunks);
int rsmp_fft_resample_bulk_device(rsmp_fft* r, const float* d_in, size_t in_len, float* d_out,
                                  size_t out_len, size_t n_chunks, void* stream);
/* n instances with the same rate pair on one device, one launch. */
int rsmp_fft_batch_resample_bulk_device(rsmp_fft* const* rs, size_t n, const float* const* d_in,
                                        float* const* d_out, const size_t* n_chunks, void* stream);
/* The same over WAV samples as they sit in the file (resample/src/main.rs:128-137): d_pcm[i] = little-endian PCM of `bits`
 * (16 / 24 / 32) per sample, two channels a frame, n_chunks[i] * chunk_size_input samples; converted inside the kernel's
 * first load (`sample as f32 / (1 << (bits - 1)) as f32`, with the reference's 32-bit divisor -2^31) -- bit for bit what
 * rsmp_pcm_to_stereo_f32_device + rsmp_fft_batch_resample_bulk_device give, one pass over HBM less.  Two-channel streams
 * of the rate pairs the wave kernel serves (72 of the 90); RSMP_ERR_INVALID_ARGUMENT otherwise.  Alignment of d_pcm[i]:
 * 8 / 4 / 16 bytes for 16 / 24 / 32 bits. */
int rsmp_fft_batch_resample_bulk_pcm_device(rsmp_fft* const* rs, size_t n, const void* const* d_pcm, int bits,
                                            float* const* d_out, const size_t* n_chunks, void* stream);

/* host-only: block sizes and the N/2-point Stockham stage lists for a rate pair
 * (planner.rs:35-245, optimizer.rs:6-64, radix_fft.rs:222-246). */
int rsmp_fft_plan_sizes(uint32_t input_rate_hz, uint32_t output_rate_hz, size_t* fft_size_input,
                        size_t* fft_size_output, int* forward_stages, size_t* n_forward_stages,
                        int* inverse_stages, size_t* n_inverse_stages, size_t max_stages);

#ifdef __cplusplus
}
#endif
#endif
