//! Drop-in replacement for the public API of hasenbanck/resampler v0.5.1 (`ResamplerFir`,
//! `ResamplerFft`, `SampleRate`, `Latency`, `Attenuation`, `ResampleError`; reference
//! src/lib.rs:160-188) on top of the C ABI of libresampler_amd.so (include/resampler_amd.h).
//!
//! SOURCE ONLY -- never compiled in the build image (no rustc).  Same names, argument meaning and
//! error behaviour as the reference; where the reference panics (zero rates) this shim panics too.
use std::os::raw::{c_int, c_void};

#[derive(Copy, Clone, Hash, PartialEq, Eq, PartialOrd, Ord, Debug)]
pub enum ResampleError {
    InvalidInputBufferSize,
    InvalidOutputBufferSize,
}

impl core::fmt::Display for ResampleError {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
        match self {
            Self::InvalidInputBufferSize => f.write_str("Input buffer size is invalid"),
            Self::InvalidOutputBufferSize => f.write_str("Output buffer size is invalid"),
        }
    }
}
impl std::error::Error for ResampleError {}

/// Same variant order as the reference (src/lib.rs:167-188) == RSMP_HZ* of the C ABI.
#[repr(i32)]
#[derive(Debug, Copy, Clone, PartialOrd, PartialEq, Ord, Eq, Hash)]
pub enum SampleRate {
    Hz22050 = 0, Hz16000, Hz32000, Hz44100, Hz48000, Hz88200, Hz96000, Hz176400, Hz192000, Hz384000,
}

const ALL_RATES: [SampleRate; 10] = [
    SampleRate::Hz22050, SampleRate::Hz16000, SampleRate::Hz32000, SampleRate::Hz44100, SampleRate::Hz48000,
    SampleRate::Hz88200, SampleRate::Hz96000, SampleRate::Hz176400, SampleRate::Hz192000, SampleRate::Hz384000,
];

/// impl From<SampleRate> for u32 (src/lib.rs:219-236): the table lives in the library
/// (rsmp_sample_rate_hz), so the shim cannot drift from it.
impl From<SampleRate> for u32 {
    fn from(value: SampleRate) -> Self { unsafe { rsmp_sample_rate_hz(value as c_int) } }
}

/// impl TryFrom<u32> for SampleRate (src/lib.rs:238-257): `Err(())` for a rate that is not in the enum.
impl TryFrom<u32> for SampleRate {
    type Error = ();
    fn try_from(value: u32) -> Result<Self, Self::Error> {
        ALL_RATES.iter().copied().find(|r| u32::from(*r) == value).ok_or(())
    }
}

#[repr(i32)]
#[derive(Default, Debug, Copy, Clone, PartialEq, Eq, Hash)]
pub enum Latency { Sample8 = 0, Sample16, Sample32, #[default] Sample64 }

#[repr(i32)]
#[derive(Default, Debug, Copy, Clone, PartialEq, Eq, Hash)]
pub enum Attenuation { Db60 = 0, Db90, #[default] Db120 }

#[allow(non_camel_case_types)]
type rsmp_fir = c_void;
#[allow(non_camel_case_types)]
type rsmp_fft = c_void;
#[allow(non_camel_case_types)]
type rsmp_fir_lockstep = c_void;

extern "C" {
    fn rsmp_fir_new(channels: usize, input_rate: c_int, output_rate: c_int, latency: c_int,
                    attenuation: c_int, device: c_int) -> *mut rsmp_fir;
    fn rsmp_fir_new_from_hz(channels: usize, in_hz: u32, out_hz: u32, latency: c_int,
                            attenuation: c_int, device: c_int) -> *mut rsmp_fir;
    fn rsmp_fir_free(r: *mut rsmp_fir);
    fn rsmp_fir_buffer_size_output(r: *const rsmp_fir) -> usize;
    fn rsmp_fir_delay(r: *const rsmp_fir) -> usize;
    fn rsmp_fir_reset(r: *mut rsmp_fir);
    fn rsmp_fir_resample(r: *mut rsmp_fir, input: *const f32, in_len: usize, output: *mut f32,
                         out_len: usize, consumed: *mut usize, produced: *mut usize) -> c_int;
    fn rsmp_fft_new(channels: usize, input_rate: c_int, output_rate: c_int, device: c_int) -> *mut rsmp_fft;
    fn rsmp_fft_free(r: *mut rsmp_fft);
    fn rsmp_fft_chunk_size_input(r: *const rsmp_fft) -> usize;
    fn rsmp_fft_chunk_size_output(r: *const rsmp_fft) -> usize;
    fn rsmp_fft_delay(r: *const rsmp_fft) -> usize;
    fn rsmp_fft_resample(r: *mut rsmp_fft, input: *const f32, in_len: usize, output: *mut f32,
                         out_len: usize) -> c_int;
    fn rsmp_last_error() -> *const std::os::raw::c_char;
    fn rsmp_sample_rate_hz(sample_rate: c_int) -> u32;
    fn rsmp_fir_channels(r: *const rsmp_fir) -> usize;
    fn rsmp_fir_taps(r: *const rsmp_fir) -> usize;
    fn rsmp_fir_phases(r: *const rsmp_fir) -> usize;
    fn rsmp_fft_channels(r: *const rsmp_fft) -> usize;
    // addition: WAV samples (resample/src/main.rs:128-137) converted where the FIR kernels read their input
    fn rsmp_fir_batch_resample_bulk_pcm_device(rs: *const *mut rsmp_fir, n: usize, d_pcm: *const *const std::os::raw::c_void, bits: c_int,
                                               in_lens: *const usize, chunk_len: usize, d_out: *const *mut f32, out_caps: *const usize,
                                               consumed: *mut usize, produced: *mut usize, stream: *mut std::os::raw::c_void) -> c_int;
    // addition: WAV samples (resample/src/main.rs:128-137) converted inside the FFT kernel's first load
    fn rsmp_fft_batch_resample_bulk_pcm_device(rs: *const *mut rsmp_fft, n: usize, d_pcm: *const *const std::os::raw::c_void, bits: c_int,
                                               d_out: *const *mut f32, n_chunks: *const usize, stream: *mut std::os::raw::c_void) -> c_int;
    // additions to the reference API: a fixed set of streams stepped together on device-resident state
    fn rsmp_fir_lockstep_new(rs: *const *mut rsmp_fir, n: usize, max_step_frames: usize) -> *mut rsmp_fir_lockstep;
    fn rsmp_fir_lockstep_free(ls: *mut rsmp_fir_lockstep);
    fn rsmp_fir_lockstep_bind(ls: *mut rsmp_fir_lockstep, d_in: *const *const f32, d_out: *const *mut f32,
                              out_caps: *const usize) -> c_int;
    fn rsmp_fir_lockstep_step(ls: *mut rsmp_fir_lockstep, in_frames: usize, in_offset_frames: usize,
                              d_in_frames: *const u32, append: c_int, stream: *mut c_void) -> c_int;
    fn rsmp_fir_lockstep_run(ls: *mut rsmp_fir_lockstep, k_steps: usize, in_frames: usize, in_offset_frames: usize,
                             append: c_int, stream: *mut c_void) -> c_int;
    fn rsmp_fir_lockstep_counts(ls: *mut rsmp_fir_lockstep, consumed: *mut usize, produced: *mut usize) -> c_int;
    fn rsmp_fir_lockstep_run_counts(ls: *mut rsmp_fir_lockstep, consumed: *mut usize, produced: *mut usize,
                                    max_steps: usize) -> c_int;
    fn rsmp_fir_lockstep_sync(ls: *mut rsmp_fir_lockstep) -> c_int;
    fn rsmp_fir_lockstep_sync_totals(ls: *mut rsmp_fir_lockstep, accepted: *mut usize, produced: *mut usize, status_or: *mut u32) -> c_int;
    #[allow(dead_code)]
    fn rsmp_fir_lockstep_in_sync(ls: *const rsmp_fir_lockstep, in_sync: *mut c_int) -> c_int;
    #[allow(dead_code)]
    fn rsmp_fir_lockstep_discard(ls: *mut rsmp_fir_lockstep);
    #[allow(dead_code)]
    fn rsmp_fir_lockstep_rebind_buffers(ls: *mut rsmp_fir_lockstep, d_in: *const *const f32, d_out: *const *mut f32, stream: *mut c_void) -> c_int;
    #[allow(dead_code)]
    fn rsmp_fir_batch_distinct_states(rs: *const *mut rsmp_fir, n: usize, distinct: *mut usize) -> c_int;
    fn rsmp_fir_lockstep_table_rebinds(ls: *const rsmp_fir_lockstep, rebinds: *mut usize) -> c_int;
    fn rsmp_fir_lockstep_run_bulk(ls: *mut rsmp_fir_lockstep, total_frames: usize, chunk_frames: usize, in_offset_frames: usize,
                                  append: c_int, stream: *mut std::os::raw::c_void) -> c_int;
    fn rsmp_fir_lockstep_stats(ls: *const rsmp_fir_lockstep, out: *mut u64, n: usize) -> c_int;
    fn rsmp_fir_lockstep_set_drift_policy(ls: *mut rsmp_fir_lockstep, tolerance_frames: f64, check_frames: usize) -> c_int;
}

fn device() -> c_int {
    std::env::var("RESAMPLER_AMD_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0)
}

fn last_error() -> String {
    unsafe { std::ffi::CStr::from_ptr(rsmp_last_error()).to_string_lossy().into_owned() }
}

fn status(rc: c_int) -> Result<(), ResampleError> {
    match rc {
        0 => Ok(()),
        1 => Err(ResampleError::InvalidInputBufferSize),
        2 => Err(ResampleError::InvalidOutputBufferSize),
        _ => panic!("resampler_amd: {}", last_error()),
    }
}

/// src/resampler_fir.rs:179-643
pub struct ResamplerFir { handle: *mut rsmp_fir }
unsafe impl Send for ResamplerFir {}
unsafe impl Sync for ResamplerFir {}

impl ResamplerFir {
    pub fn new(channels: usize, input_rate: SampleRate, output_rate: SampleRate, latency: Latency,
               attenuation: Attenuation) -> Self {
        let handle = unsafe {
            rsmp_fir_new(channels, input_rate as c_int, output_rate as c_int, latency as c_int,
                         attenuation as c_int, device())
        };
        assert!(!handle.is_null(), "{}", last_error());
        Self { handle }
    }

    pub fn new_from_hz(channels: usize, input_rate_hz: u32, output_rate_hz: u32, latency: Latency,
                       attenuation: Attenuation) -> Self {
        assert!(input_rate_hz > 0, "input sample rate must be greater than zero");
        assert!(output_rate_hz > 0, "output sample rate must be greater than zero");
        let handle = unsafe {
            rsmp_fir_new_from_hz(channels, input_rate_hz, output_rate_hz, latency as c_int,
                                 attenuation as c_int, device())
        };
        assert!(!handle.is_null(), "{}", last_error());
        Self { handle }
    }

    pub fn buffer_size_output(&self) -> usize { unsafe { rsmp_fir_buffer_size_output(self.handle) } }
    pub fn delay(&self) -> usize { unsafe { rsmp_fir_delay(self.handle) } }
    pub fn reset(&mut self) { unsafe { rsmp_fir_reset(self.handle) } }

    pub fn resample(&mut self, input: &[f32], output: &mut [f32]) -> Result<(usize, usize), ResampleError> {
        let (mut consumed, mut produced) = (0usize, 0usize);
        let rc = unsafe {
            rsmp_fir_resample(self.handle, input.as_ptr(), input.len(), output.as_mut_ptr(),
                              output.len(), &mut consumed, &mut produced)
        };
        status(rc).map(|_| (consumed, produced))
    }

    /// Addition to the reference API: the driver loop of resample/src/main.rs:226-254 (calls of `chunk_len` samples) over
    /// a two-channel WAV file's samples as they are in the file -- `n_samples` little-endian PCM samples of `bits`
    /// (16 / 24 / 32) in device memory -- into `d_out` (device memory, `out_cap` values); the conversion of
    /// main.rs:128-137 happens where the kernels read their input.  Returns (consumed, produced) in samples.
    pub unsafe fn resample_bulk_pcm_device(&mut self, d_pcm: *const std::os::raw::c_void, bits: u32, n_samples: usize, chunk_len: usize,
                                           d_out: *mut f32, out_cap: usize) -> Result<(usize, usize), ResampleError> {
        let (h, p, o) = ([self.handle], [d_pcm], [d_out]);
        let (mut consumed, mut produced) = (0usize, 0usize);
        status(rsmp_fir_batch_resample_bulk_pcm_device(h.as_ptr(), 1, p.as_ptr(), bits as c_int, &n_samples, chunk_len, o.as_ptr(), &out_cap,
                                                       &mut consumed, &mut produced, std::ptr::null_mut())).map(|_| (consumed, produced))
    }
}

/// impl fmt::Debug for ResamplerFir (src/resampler_fir.rs:203-211): channels, taps, phases, `..`
impl core::fmt::Debug for ResamplerFir {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
        let (channels, taps, phases) = unsafe {
            (rsmp_fir_channels(self.handle), rsmp_fir_taps(self.handle), rsmp_fir_phases(self.handle))
        };
        f.debug_struct("ResamplerFir")
            .field("channels", &channels)
            .field("taps", &taps)
            .field("phases", &phases)
            .finish_non_exhaustive()
    }
}

impl Drop for ResamplerFir {
    fn drop(&mut self) { unsafe { rsmp_fir_free(self.handle) } }
}

/// src/resampler_fft.rs:43-240
pub struct ResamplerFft { handle: *mut rsmp_fft }
unsafe impl Send for ResamplerFft {}
unsafe impl Sync for ResamplerFft {}

impl ResamplerFft {
    pub fn new(channels: usize, sample_rate_input: SampleRate, sample_rate_output: SampleRate) -> Self {
        let handle = unsafe {
            rsmp_fft_new(channels, sample_rate_input as c_int, sample_rate_output as c_int, device())
        };
        assert!(!handle.is_null(), "{}", last_error());
        Self { handle }
    }
    pub fn chunk_size_input(&self) -> usize { unsafe { rsmp_fft_chunk_size_input(self.handle) } }
    pub fn chunk_size_output(&self) -> usize { unsafe { rsmp_fft_chunk_size_output(self.handle) } }
    pub fn delay(&self) -> usize { unsafe { rsmp_fft_delay(self.handle) } }
    /// `n_chunks` chunks of a two-channel WAV file's samples as they are in the file -- little-endian PCM of `bits`
    /// (16 / 24 / 32) per sample in device memory -- resampled into `d_out` (device memory, n_chunks * chunk_size_output()
    /// values): the conversion of resample/src/main.rs:128-137 happens inside the kernel's first load.
    pub unsafe fn resample_bulk_pcm_device(&mut self, d_pcm: *const std::os::raw::c_void, bits: u32, d_out: *mut f32, n_chunks: usize) -> Result<(), ResampleError> {
        let (h, p, o) = ([self.handle], [d_pcm], [d_out]);
        status(rsmp_fft_batch_resample_bulk_pcm_device(h.as_ptr(), 1, p.as_ptr(), bits as c_int, o.as_ptr(), &n_chunks, std::ptr::null_mut()))
    }
    pub fn resample(&mut self, input: &[f32], output: &mut [f32]) -> Result<(), ResampleError> {
        status(unsafe {
            rsmp_fft_resample(self.handle, input.as_ptr(), input.len(), output.as_mut_ptr(), output.len())
        })
    }
}

/// impl fmt::Debug for ResamplerFft (src/resampler_fft.rs:56-66)
impl core::fmt::Debug for ResamplerFft {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
        let channels = unsafe { rsmp_fft_channels(self.handle) };
        f.debug_struct("ResamplerFft")
            .field("channels", &channels)
            .field("chunk_size_input", &self.chunk_size_input())
            .field("chunk_size_output", &self.chunk_size_output())
            .field("fft_size_input", &(self.chunk_size_input() / channels))
            .field("fft_size_output", &(self.chunk_size_output() / channels))
            .finish_non_exhaustive()
    }
}

impl Drop for ResamplerFft {
    fn drop(&mut self) { unsafe { rsmp_fft_free(self.handle) } }
}

/// NOT part of the reference API: a fixed set of `ResamplerFir` streams fed one chunk each per step from buffers in
/// GPU memory (include/resampler_amd.h, rsmp_fir_lockstep_*).  `step` is one `resample()` call per stream
/// (src/resampler_fir.rs:509-621), `run` is k of them per stream in one go -- the loop of resample/src/main.rs:226-254
/// over a resident buffer -- planned and computed on the device.  Pointers are device pointers (hipMalloc).
pub struct LockstepBatch { handle: *mut rsmp_fir_lockstep, streams: Vec<ResamplerFir>, last_run: usize }
unsafe impl Send for LockstepBatch {}

impl LockstepBatch {
    pub fn new(streams: Vec<ResamplerFir>, max_step_frames: usize) -> Self {
        let handles: Vec<*mut rsmp_fir> = streams.iter().map(|s| s.handle).collect();
        let handle = unsafe { rsmp_fir_lockstep_new(handles.as_ptr(), handles.len(), max_step_frames) };
        assert!(!handle.is_null(), "{}", last_error());
        Self { handle, streams, last_run: 0 }
    }
    /// `out_caps[i]` >= `buffer_size_output()` of stream i: the room of ONE call; the buffers hold a whole run.
    pub unsafe fn bind(&mut self, d_in: &[*const f32], d_out: &[*mut f32], out_caps: &[usize]) -> Result<(), ResampleError> {
        assert!(d_in.len() == self.streams.len() && d_out.len() == d_in.len() && out_caps.len() == d_in.len());
        status(rsmp_fir_lockstep_bind(self.handle, d_in.as_ptr(), d_out.as_ptr(), out_caps.as_ptr()))
    }
    pub fn step(&mut self, in_frames: usize, in_offset_frames: usize, append: bool) -> Result<(), ResampleError> {
        self.last_run = 0;   // (a step is not a run: run_counts() has nothing to report, counts() has the step's)
        status(unsafe { rsmp_fir_lockstep_step(self.handle, in_frames, in_offset_frames, std::ptr::null(), append as c_int, std::ptr::null_mut()) })
    }
    pub fn run(&mut self, k_steps: usize, in_frames: usize, in_offset_frames: usize, append: bool) -> Result<(), ResampleError> {
        status(unsafe { rsmp_fir_lockstep_run(self.handle, k_steps, in_frames, in_offset_frames, append as c_int, std::ptr::null_mut()) })?;
        self.last_run = k_steps;
        Ok(())
    }
    /// A whole buffer per stream in calls of `chunk_frames` frames, the last one shorter (the loop of
    /// resample/src/main.rs:226-254), planned on the device whatever states the streams are in.
    pub fn run_bulk(&mut self, total_frames: usize, chunk_frames: usize, in_offset_frames: usize, append: bool) -> Result<(), ResampleError> {
        // (the C side answers chunk_frames == 0 with RSMP_ERR_INVALID_ARGUMENT; last_run moves on success only)
        status(unsafe { rsmp_fir_lockstep_run_bulk(self.handle, total_frames, chunk_frames, in_offset_frames, append as c_int, std::ptr::null_mut()) })?;
        self.last_run = total_frames / chunk_frames;
        Ok(())
    }
    /// (consumed, produced) of the last call of every stream, in f32 values; waits for the launch.
    pub fn counts(&mut self) -> Vec<(usize, usize)> {
        let n = self.streams.len();
        let (mut c, mut p) = (vec![0usize; n], vec![0usize; n]);
        status(unsafe { rsmp_fir_lockstep_counts(self.handle, c.as_mut_ptr(), p.as_mut_ptr()) }).expect("counts");
        c.into_iter().zip(p).collect()
    }
    /// ... of every call of the last run: `[call][stream]`.
    pub fn run_counts(&mut self) -> Vec<Vec<(usize, usize)>> {
        let (n, k) = (self.streams.len(), self.last_run);
        if k == 0 { return Vec::new(); }   // no run yet, or a step since (the C side answers that with an error)
        let (mut c, mut p) = (vec![0usize; n * k], vec![0usize; n * k]);
        status(unsafe { rsmp_fir_lockstep_run_counts(self.handle, c.as_mut_ptr(), p.as_mut_ptr(), k) }).expect("run_counts");
        (0..k).map(|s| (0..n).map(|i| (c[s * n + i], p[s * n + i])).collect()).collect()
    }
    /// Times the batch replaced a class of streams' coefficient tables because their f64 position drift had moved on
    /// (diagnostic; the batch watches the drift itself).
    pub fn table_rebinds(&self) -> usize {
        let mut v = 0usize;
        status(unsafe { rsmp_fir_lockstep_table_rebinds(self.handle, &mut v) }).expect("table_rebinds");
        v
    }
    /// Diagnostic counters (include/resampler_amd.h, rsmp_fir_lockstep_stats): table rebinds, runs taken over from the
    /// plan stream, runs planned ahead and dropped, late table polls, waits for the table worker, plan-stream probes,
    /// whether the last run's stream has a plan stream, drift classes.
    pub fn stats(&self) -> [u64; 8] {
        let mut v = [0u64; 8];
        status(unsafe { rsmp_fir_lockstep_stats(self.handle, v.as_mut_ptr(), v.len()) }).expect("stats");
        v
    }
    /// How closely the coefficient tables follow the streams' f64 drift (default 1.2e-7 of a frame, looked at every 2^19 frames).
    pub fn set_drift_policy(&mut self, tolerance_frames: f64, check_frames: usize) -> Result<(), ResampleError> {
        status(unsafe { rsmp_fir_lockstep_set_drift_policy(self.handle, tolerance_frames, check_frames) })
    }
    /// Waits for what has been launched, writes the device state back into the streams and returns, per stream, the values accepted and
    /// produced since the states were last exchanged (what a bulk driver loop over the same input returns) and the OR of the status flags.
    pub fn sync_totals(&mut self) -> Result<(Vec<usize>, Vec<usize>, u32), ResampleError> {
        let n = self.streams.len();
        let (mut a, mut p, mut f) = (vec![0usize; n], vec![0usize; n], 0u32);
        status(unsafe { rsmp_fir_lockstep_sync_totals(self.handle, a.as_mut_ptr(), p.as_mut_ptr(), &mut f) })?;
        Ok((a, p, f))
    }
    /// Writes the device state back into the streams and hands them back.
    pub fn into_streams(mut self) -> Vec<ResamplerFir> {
        unsafe { rsmp_fir_lockstep_sync(self.handle); }
        std::mem::take(&mut self.streams)
    }
}

impl Drop for LockstepBatch {
    fn drop(&mut self) { unsafe { rsmp_fir_lockstep_free(self.handle) } }
}
