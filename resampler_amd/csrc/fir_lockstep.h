// fir_lockstep.h -- lock-step batch of ResamplerFir streams with DEVICE-resident state.
//
// BASELINE config 4 (a thousand independent streams of mixed rate pairs, each fed one 512-frame
// chunk per step) is microseconds of GPU work per step: any per-stream host work -- replaying the
// state machine, assembling descriptors, uploading them -- costs more than the arithmetic.  A lock-step
// batch therefore keeps everything a step needs in HBM: the streams' reference state
// (read_position / available_frames / position, resampler_fir.rs:189-192), their buffered frames, their
// buffer pointers.  One step is ONE kernel launch with constant arguments; the kernel runs the
// reference's control flow itself (fir_mirror_core.h, one lane per stream), stages every stream's
// [buffered | new] frames in LDS, computes the outputs on the matrix cores with "row = stream"
// (the 16 columns of a tile are (stream, period) pairs, so short steps of many streams fill the
// tiles), retires the consumed frames and writes (consumed, produced) per stream to HBM.
// Two-channel streams use the arithmetic of fir_split.hip (operands cut into two fp16 planes, three
// v_mfma_f32_16x16x32_f16 products per term, f32 accumulation): 5x less matrix-pipe time than exact f32
// products, which stay available (rsmp_fir_set_kernel(.., RSMP_FIR_KERNEL_PERIODIC_F32), other channel counts).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "fir_mirror_core.h"
#include "fir_mirror_fast.h"
#include "fir_kernels.h"
#include "fir_periodic.h"

namespace rsmp {

constexpr uint32_t kLsWaves = 8;              // waves per workgroup (two workgroups per CU: <= 128 VGPRs)
constexpr uint32_t kLsMaxSlots = 16;          // streams per workgroup
constexpr uint32_t kLsSegCap = 40;            // exact position runs kept per stream and step
constexpr uint32_t kLsMaxBlk = 12;            // 16-tap blocks of a tile window held in registers (row_len <= 192)
constexpr uint32_t kLsLdsLimit = 160 * 1024;

struct LockstepStream {        // per bound stream, constant between binds (HBM)
    const float* in;           // a step reads in + in_offset, in_frames frames
    float* out;
    float* hist;               // frames buffered between steps (interleaved): a step with an even index reads
    float* hist_alt;           // `hist` and leaves its tail in `hist_alt`, an odd one the other way round
    const float* coeffs;       // [1024][taps] polyphase table
    uint64_t out_cap_frames;   // room of `out` per step, in frames
};

struct LockstepGroup {         // one workgroup's share: `count` streams of one geometry (HBM)
    uint32_t first, count;     // streams [first, first + count) of the batch's internal order
    uint32_t channels, taps;
    uint32_t periodic;         // 0: every output in the reference's two-row form (any ratio)
    uint32_t num, den;         // in_hz / out_hz reduced
    uint32_t a, b;             // super period: a = r * num input frames -> b = r * den outputs
    uint32_t row_len, n_tiles; // padded window of a 16-class tile; tiles per super period
    uint32_t guard_frames;     // zeroed frames in front of a stream's span in LDS (>= a)
    uint32_t span_frames;      // capacity of the span itself (buffered + new frames)
    uint32_t region_frames;    // guard + span + zeroed tail (>= a + row_len)
    uint32_t max_out;          // output frames one step can produce
    uint32_t wrap_words;       // bitmap words per stream: ceil(max_out / 32)
    uint32_t wrap_cap;         // wrap list entries per stream
    uint32_t max_cols;         // column table entries
    const float* class_coef;   // [tile][row_len / 16][64 lanes][4 steps] (A-operand order)
    const TileMeta* class_meta;
    uint32_t lds_bytes;        // what this group needs
    uint32_t slots;            // streams per workgroup the LDS layout is sized for (>= count)
    uint32_t split;            // 1: two-channel streams on the fp16 matrix cores with split operands (fir_split.hip's
                               //    arithmetic): class_coef is the split table, the LDS holds a transposed fp16 image
    uint32_t rows;             // split: rows (frames) of the image: last tile's window start + row_len
    uint32_t row_bytes;        // split: bytes per image row: 160 (32 B of padding: conflict-free transposed reads), or
                               //        128 where only the unpadded image leaves room for two workgroups per CU
    uint32_t pad0;
};

struct LockstepArgs {
    const LockstepGroup* groups;
    const LockstepStream* streams;
    FirMirrorState* states;
    uint64_t* out_cursor;                 // f32 values appended so far per stream (append mode)
    uint64_t* counts;                     // [n][2]: (consumed, produced) of the step, in f32 values
    uint32_t* status;                     // sticky per-stream flags (kLsStatus*)
    const uint32_t* order;                // internal stream index -> index in the caller's order
    const uint32_t* in_frames_per_stream; // optional: frames offered per stream, in the caller's order
    uint64_t in_offset;                   // frames added to every stream's `in`
    uint32_t in_frames;                   // frames offered to every stream (when the array is null)
    uint32_t append;                      // 1: a step's output goes to out + out_cursor; 0: to out
    uint32_t in_aligned8;                 // every two-channel stream's `in` is 8-byte aligned (split variant: 8-byte loads)
    unsigned long long* trace;            // diagnostic instantiation only (RSMP_LS_TRACE), else null
    // Plans are computed one step AHEAD: while the other waves of a workgroup compute step k, its first
    // wave runs the state machine for step k + 1 (assuming the same number of frames will be offered) and
    // leaves the result in the stream's plan record; step k + 1 then starts from the record instead of
    // ~20 k cycles of serial f64 arithmetic.  A record is used only if its epoch, step and frame count
    // match; otherwise the step plans in line as before.
    uint32_t* peaks;                      // [n_streams][4]: epoch, step it is a prediction for, bits of the stream's latest peak, -
    char* recs;                           // [2][n_streams] records of rec_stride bytes (parity = step & 1)
    uint32_t rec_stride, n_streams;
    uint32_t epoch, step;
    uint32_t hist_parity;                 // 0: this step reads `hist` and leaves its tail in `hist_alt`; 1: the other way round
};

struct LsPlanHeader {          // head of a plan record; followed by kLsSegCap runs (24 B each), then the wrap list
    uint32_t epoch, step, in_frames, n_out;
    uint32_t hist_frames, accepted, consumed, tail_frames;
    uint32_t n_segs, n_wraps, flags, pad0;
    uint64_t abs_out, abs_consumed;
    FirMirrorState after;      // the stream's state after the step
    uint64_t pad1;
};
static_assert(sizeof(LsPlanHeader) == 160, "plan record layout");
constexpr uint32_t kLsRecSegs = 160, kLsRecWraps = kLsRecSegs + kLsSegCap * 24;
inline uint32_t lockstep_rec_stride(uint32_t wrap_cap) { return (kLsRecWraps + 4 * wrap_cap + 15) / 16 * 16; }

constexpr uint32_t kLsStatusRunOverflow = 1;   // more than kLsSegCap position runs in one step
constexpr uint32_t kLsStatusNonFinite = 2;     // a step saw non-finite samples (reference-form path taken)
constexpr uint32_t kLsStatusAperiodic = 4;     // the f64 drift left the class tables' tolerance
constexpr uint32_t kLsStatusPartialAccept = 8; // rsmp_fir_lockstep_run: a call accepted fewer frames than it was offered
constexpr uint32_t kLsStatusPlannerCheck = 16; // rsmp_fir_lockstep_run: the replay of a call found a premise of the planner's closed form violated (never observed)

// Geometry of one (rate pair, taps, channels, step size) combination.
struct LockstepGeometry {
    bool periodic = false;
    uint32_t num = 0, den = 0, r = 0, a = 0, b = 0, taps = 0, row_len = 0, n_tiles = 0;
    uint32_t guard_frames = 0, span_frames = 0, region_frames = 0, max_out = 0, cols_per_stream = 0;
    uint32_t slots = 1;        // streams per workgroup
    uint32_t wrap_words = 0, wrap_cap = 0, max_cols = 0, lds_bytes = 0;
    bool split = false;        // fp16x2 split operands (two-channel streams, unless exact f32 products are asked for)
    uint32_t rows = 0;         // split: rows of the LDS image
    uint32_t row_bytes = 0;    // split: bytes per image row (160, or 128 without padding)
};
// allow_split = false: exact-f32 products (RSMP_FIR_KERNEL_PERIODIC_F32 on the streams, or RSMP_LS_EXACT=1).
LockstepGeometry lockstep_geometry(uint64_t num, uint64_t den, double ratio, uint32_t taps,
                                   uint32_t channels, uint32_t step_frames, bool allow_split = true);
// The PeriodicGeometry view of it that build_class_table understands (f32 matrix-core layout, or the split
// kernel's fp16x2 layout).
PeriodicGeometry lockstep_class_geometry(const LockstepGeometry& g);
constexpr uint32_t kLsImageRowBytes = 160;   // split image: (2 channels x 2 planes) x 32 B + 32 B of padding (fir_split.hip)
constexpr uint32_t kLsImageRowBytesPacked = 128;   // ... without the padding
constexpr uint32_t kLsLdsPerWorkgroup = 80 * 1024 - 512;   // two workgroups per CU (160 KB, less the allocation granule)

hipError_t launch_fir_lockstep(const LockstepArgs& args, uint32_t n_groups, uint32_t max_lds_bytes,
                               hipStream_t stream);

// ---- rsmp_fir_lockstep_run: k calls per stream, planned on the device, computed by the bulk kernels ------------
struct LsRunStream {           // per stream (internal order), constant for the batch
    uint32_t wrap_unit;        // the bitmap of wrapped outputs starts at output (abs_out / wrap_unit) * wrap_unit: the bulk
                               // geometry's b (= den unless a super period of an exact ratio, whose outputs never wrap)
    uint32_t den, channels;
    uint32_t caller;           // the stream's index in the caller's order
    // the class tables of the bulk kernels and the drift they were built for: replaced as the stream's f64 drift moves on
    // (fir_lockstep_api.cpp, DriftClass); the planner copies them into every run's descriptor
    const float* class_coef;
    const float* class_wrap_coef;
    const TileMeta* class_meta;
    double drift;
};
struct LsRunArgs {
    const LockstepStream* streams;
    const LsRunStream* rs;
    const FirMirrorState* states_in;
    FirMirrorState* states_out;
    FirMirrorState* states_before;   // [n]: copy of the states the run started from
    MirrorPred* preds;         // [n][k]: the predicted structure of every call (fir_mirror_fast.h)
    void* call_recs;           // [n][k] x 24 bytes: what the chain leaves per call for the replay
    const uint64_t* cursor_in;
    uint64_t* cursor_out;
    FirStreamDesc* descs;      // [n]: the run's descriptors (the constant fields are the host's)
    uint32_t* wrap_bits;       // [n][wrap_words]
    uint32_t* counts;          // [k][n][2] in the caller's order: (consumed, produced) per call in f32 values
    uint64_t* last_counts;     // [n][2], internal order: the last call's (rsmp_fir_lockstep_counts)
    uint32_t* status;          // per-stream flags: ORed into
    uint32_t* zero_status;     // non-null: `status` is a scratch copy of a run planned AHEAD -- cleared here first (the commit ORs it in)
    uint64_t in_offset;        // frames added to every stream's `in`
    uint32_t n_streams, k, in_frames, wrap_words, append, hist_parity;
    uint32_t parallel_chain;   // K2: chunks of lean calls by the parallel chain (set by launch_fir_lockstep_plan; RSMP_LS_PCHAIN=0, debug: never)
};
struct LsCommitArgs {
    FirMirrorState* states; const FirMirrorState* sp_states;
    uint64_t* cursor; const uint64_t* sp_cursor;
    uint64_t* last_counts; const uint64_t* sp_last_counts;
    uint32_t* status; const uint32_t* sp_status;
    uint32_t n_streams;
};
// parts: 1 = K1 (the predictions), 2 = K2 + K3 (chain, replay); 3 = all three in `stream`
// (commit, with part 1: K1 also does what launch_fir_lockstep_commit does -- the first thread of a stream's calls copies the
// stream's scratch results into place -- and reads the states it predicts from out of the scratch copies: one launch less in
// front of a run planned ahead)
// The planner's serial kernels (chain, replay) pack kLsPlanPack streams into a workgroup -- one CU -- for batches of fewer than
// kLsPlanPackBelow streams: the CUs they take are then few and known (lockstep_plan_cus), whoever reaches the chip first.
constexpr uint32_t kLsPlanPack = 4, kLsPlanPackBelow = 256;   // (pack 1 / 2 / 4 / 8 at 128 streams: 0.89 / 0.71-0.86 / 0.73 / 0.83-0.95 us per step, profiles/r06/ab_c4_shard.txt: eight waves of this much CODE on one CU starve each other of instructions)
uint32_t lockstep_plan_pack(size_t n_streams);   // (fir_lockstep_run.hip; RSMP_LS_PACK, debug: 1 / 2 / 4 / 8)
// CUs the replay (K3) of a run of k calls takes when it has a wave per chunk (batches below kLsPlanPackBelow streams): sixteen waves a CU
uint32_t lockstep_replay_cus(size_t n_streams, uint32_t k);   // (fir_lockstep_run.hip)
inline uint32_t lockstep_plan_cus(size_t n_streams) {
    const uint32_t pack = lockstep_plan_pack(n_streams);
    return pack > 1 ? static_cast<uint32_t>((n_streams + pack - 1) / pack) : static_cast<uint32_t>((n_streams + 3) / 4);
}
// (k1_done, with part 1: an event the K1 launch itself completes -- hipExtLaunchKernel's stop event --, no packet of its own
// behind it as hipEventRecord would put there)
hipError_t launch_fir_lockstep_plan(const LsRunArgs& args, hipStream_t stream, int parts = 3, const LsCommitArgs* commit = nullptr,
                                    hipEvent_t k1_done = nullptr);
// out[c] = states[reps[c]].drift: the drifts the batch's classes are watched by (one thread per class).
hipError_t launch_fir_lockstep_gather_drift(const FirMirrorState* states, const uint32_t* reps, double* out, uint32_t n,
                                            hipStream_t stream);
// Do two streams run side by side?  (HIP deals a handful of hardware queues to its streams in turn; two streams on one
// queue run their kernels strictly one after the other.)  `wait` goes to the one stream: a single wave that polls *flag
// until it reads `token` or `timeout_ticks` of the 100 MHz clock have passed, and stores 1 / 0 to *result (mapped host
// memory).  `set` goes to the other stream and stores `token` to *flag.  On one queue `set` cannot start before `wait`
// has given up: the result is decided by the DEVICE, whatever the host's threads are doing meanwhile (round 4 timed the
// pair with the host's clock against a 200 us threshold, which a busy host fails both ways).
hipError_t launch_fir_lockstep_probe_wait(uint32_t* flag, uint32_t token, uint32_t timeout_ticks, uint32_t* result, hipStream_t stream);
hipError_t launch_fir_lockstep_probe_set(uint32_t* flag, uint32_t token, hipStream_t stream);
// New class tables for up to kLsMaxPatches drift classes, written into the device copies of the group table (step
// kernel: the groups whose `pad0` names the class) and the run planner's stream table (streams [first, first + count))
// by ONE small kernel in stream order -- no copy-engine operation, no staging buffer to wait for.
constexpr uint32_t kLsMaxPatches = 8;
struct LsTablePatch {
    uint32_t cls, first, count, flags;     // flags: 1 = step table, 2 = run table
    const float* step_coef; const TileMeta* step_meta;
    const float* run_coef; const float* run_wrap_coef; const TileMeta* run_meta;
    double drift;
};
struct LsPatchArgs {
    LockstepGroup* groups; LsRunStream* rs;   // (rs: null before the batch's first run)
    uint32_t n_groups, n_streams, n_patches, pad;
    LsTablePatch p[kLsMaxPatches];
};
hipError_t launch_fir_lockstep_patch_tables(const LsPatchArgs& args, hipStream_t stream);
// A run planned ahead (on a stream of its own, while the previous run computes) left its results in scratch copies:
// states, append positions, the last call's counts, status flags -> the batch's own, when the run is really asked for.
hipError_t launch_fir_lockstep_commit(const LsCommitArgs& args, hipStream_t stream);
// A run planned ahead names the buffers the batch was bound to when it was planned: descs[gs].in / .out again from the stream table as
// it is now (rsmp_fir_lockstep_rebind_buffers; a run that starts at the front of `out`: no `append`).
hipError_t launch_fir_lockstep_rebase(FirStreamDesc* descs, const LockstepStream* streams, const LsRunStream* rs, uint64_t in_offset,
                                      uint32_t n_streams, hipStream_t stream);
hipError_t launch_fir_lockstep_gather_counts(const uint64_t* last_counts, const LsRunStream* rs, uint32_t* counts, uint32_t n,
                                             hipStream_t stream);

}  // namespace rsmp
