//! Safe-ish Rust wrapper over the C ABI of the MI355X batch-render path (`include/srack_hip.h`).
//!
//! NOT COMPILED in the build image (no rustc there) — reviewed against the header by hand.  It is the
//! "Rust host" of the north star: the s-rack crate would depend on this crate behind a `gpu` feature and
//! call [`Patch::execute_batch`] where the audio callback calls `synth::execute(&plan)` (src/main.rs:59-63).
//!
//! Error mapping: every C entry point returns `int` (0 ok, < 0 error).  The reference's `Result<_, ()>`
//! becomes `Result<_, Error>` with the status code and the library's thread-local message.
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct SrackPatch {
    _private: [u8; 0],
}

#[allow(non_camel_case_types)]
mod ffi {
    use super::*;
    /// `srack_kernel_cache_info` (include/srack_hip.h)
    #[repr(C)]
    pub struct SrackKernelCacheInfo {
        pub compiled: u64,
        pub disk_hits: u64,
        pub memory_hits: u64,
        pub modules_loaded: u64,
        pub code_evictions: u64,
        pub module_evictions: u64,
        pub resident_code_objects: u64,
        pub resident_modules: u64,
        pub compile_ms: f64,
        pub directory: [c_char; 512],
    }

    extern "C" {
        pub fn srack_abi_version() -> c_int;
        pub fn srack_last_error() -> *const c_char;
        pub fn srack_patch_create(sample_rate: u32, buffer_size: u32, channels: u32, out: *mut *mut SrackPatch) -> c_int;
        pub fn srack_patch_destroy(p: *mut SrackPatch) -> c_int;
        pub fn srack_patch_add_module(p: *mut SrackPatch, module_type: c_int) -> c_int;
        pub fn srack_patch_set_field(p: *mut SrackPatch, module: c_int, field: c_int, value: f64) -> c_int;
        pub fn srack_patch_get_field(p: *const SrackPatch, module: c_int, field: c_int, value: *mut f64) -> c_int;
        pub fn srack_patch_set_step(p: *mut SrackPatch, module: c_int, channel: c_int, step: c_int, state: c_int, value: c_int) -> c_int;
        pub fn srack_patch_set_output_buffer(p: *mut SrackPatch, module: c_int, port: c_int, samples: *const f32, n: u32) -> c_int;
        pub fn srack_patch_get_output_buffer(p: *const SrackPatch, module: c_int, port: c_int, dst: *mut f32, cap: u32) -> c_int;
        pub fn srack_patch_set_noise_seed(p: *mut SrackPatch, seed: u64, first_voice: u64) -> c_int;
        pub fn srack_patch_keep_state(p: *mut SrackPatch, keep: c_int) -> c_int;
        pub fn srack_patch_set_wave(p: *mut SrackPatch, module: c_int, samples: *const f32, n_samples: u32, sample_rate: f32) -> c_int;
        pub fn srack_patch_load_srk(bytes: *const c_void, n_bytes: usize, sample_rate: u32, buffer_size: u32, channels: u32, out: *mut *mut SrackPatch) -> c_int;
        pub fn srack_patch_save_srk(p: *const SrackPatch, buf: *mut c_void, cap: usize, n_bytes: *mut usize) -> c_int;
        pub fn srack_patch_connect(p: *mut SrackPatch, src: c_int, src_port: c_int, sink: c_int, sink_port: c_int) -> c_int;
        pub fn srack_patch_disconnect(p: *mut SrackPatch, sink: c_int, sink_port: c_int) -> c_int;
        pub fn srack_patch_plan(p: *mut SrackPatch, order: *mut c_int, cap: c_int) -> c_int;
        pub fn srack_voices_configure(p: *mut SrackPatch, n_voices: u32) -> c_int;
        pub fn srack_voices_set_field_f32(p: *mut SrackPatch, module: c_int, field: c_int, values: *const f32) -> c_int;
        pub fn srack_render_planes(p: *mut SrackPatch, channel_plane: *mut c_int, cap: c_int) -> c_int;
        pub fn srack_render(p: *mut SrackPatch, n_samples: u32, d_frames: *mut f32, d_mix: *mut f32, flags: u32, stream: *mut c_void) -> c_int;
        pub fn srack_device_alloc(d_ptr: *mut *mut c_void, bytes: usize) -> c_int;
        pub fn srack_device_free(d_ptr: *mut c_void) -> c_int;
        pub fn srack_device_to_host(h_dst: *mut c_void, d_src: *const c_void, bytes: usize, stream: *mut c_void) -> c_int;
        pub fn srack_device_set(device: c_int) -> c_int;
        pub fn srack_device_get(device: *mut c_int, pci_bus_id: *mut c_char, cap: usize) -> c_int;
        pub fn srack_kernel_cache_set_dir(dir: *const c_char) -> c_int;
        pub fn srack_kernel_cache_stats(out: *mut SrackKernelCacheInfo) -> c_int;
        pub fn srack_render_reserve(p: *mut SrackPatch, n_samples: u32, want_mix: c_int, flags: u32) -> c_int;
        pub fn srack_dist_unique_id(id_out: *mut u8) -> c_int;
        pub fn srack_dist_init(id: *const u8, n_ranks: c_int, rank: c_int, comm_out: *mut *mut c_void) -> c_int;
        pub fn srack_dist_comm_count(comm: *mut c_void, n_ranks: *mut c_int) -> c_int;
        pub fn srack_dist_reduce_mix(comm: *mut c_void, d_mix: *mut f32, count: usize, root: c_int, stream: *mut c_void) -> c_int;
        pub fn srack_dist_destroy(comm: *mut c_void) -> c_int;
    }
}

/// Render flags (values of `SRACK_RENDER_*`).
pub mod render_flags {
    pub const DEFAULT: u32 = 0;
    pub const EXACT_OSC: u32 = 1 << 0;
    pub const NO_FUSION: u32 = 1 << 1;
    pub const NO_UNIFORM_HOIST: u32 = 1 << 2;
    pub const NO_CTL_STAGES: u32 = 1 << 3;
    /// the general path always through the tile interpreter
    pub const NO_SPECIALIZE: u32 = 1 << 4;
    /// the general path through a kernel specialised for the program, whatever the voice count
    pub const SPECIALIZE: u32 = 1 << 5;
    /// keep the default mode's fast forms where its error bound would make oscillators (or the patch) exact: a render of seconds, not minutes
    pub const KEEP_DEFAULT: u32 = 1 << 6;
}

pub const DIST_ID_BYTES: usize = 128;

/// Module types of the hot path (values of `SRACK_MOD_*`).
#[repr(i32)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum ModuleType {
    Output = 0,
    Oscillator = 1,
    MoogFilter = 2,
    Adsr = 3,
    Vca = 4,
    MonoMixer = 5,
    Math = 6,
    GridSequencer = 7,
    PatternSequencer = 8,
    NonLinear = 9,
    Sample = 10,
    Noise = 11,
    Freeverb = 12,
}

#[derive(Debug)]
pub struct Error {
    pub code: i32,
    pub message: String,
}

fn check(rc: c_int) -> Result<c_int, Error> {
    if rc >= 0 {
        return Ok(rc);
    }
    let message = unsafe { CStr::from_ptr(ffi::srack_last_error()) }.to_string_lossy().into_owned();
    Err(Error { code: rc, message })
}

/// Device buffer owned by the caller (freed on drop).
pub struct DeviceBuffer {
    ptr: *mut c_void,
    bytes: usize,
}

impl DeviceBuffer {
    pub fn new(bytes: usize) -> Result<Self, Error> {
        let mut ptr = std::ptr::null_mut();
        check(unsafe { ffi::srack_device_alloc(&mut ptr, bytes) })?;
        Ok(Self { ptr, bytes })
    }
    pub fn as_f32(&self) -> *mut f32 {
        self.ptr as *mut f32
    }
    pub fn to_host(&self, dst: &mut [f32]) -> Result<(), Error> {
        assert!(dst.len() * 4 <= self.bytes);
        check(unsafe { ffi::srack_device_to_host(dst.as_mut_ptr() as *mut c_void, self.ptr, dst.len() * 4, std::ptr::null_mut()) }).map(|_| ())
    }
}

impl Drop for DeviceBuffer {
    fn drop(&mut self) {
        unsafe { ffi::srack_device_free(self.ptr) };
    }
}

/// The workspace's module list + plan + N voices (mirror of `SynthModuleWorkspaceImpl`, src/ui.rs:52-60).
pub struct Patch {
    raw: *mut SrackPatch,
}

impl Patch {
    /// `AudioConfig { sample_rate, buffer_size, channels }` (src/synth.rs:20-25).
    pub fn new(sample_rate: u16, buffer_size: usize, channels: u8) -> Result<Self, Error> {
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::srack_patch_create(sample_rate as u32, buffer_size as u32, channels as u32, &mut raw) })?;
        Ok(Self { raw })
    }
    /// `Module::new(&audio_config)` pushed to the module list; returns its index in `all_modules`.
    pub fn add_module(&mut self, t: ModuleType) -> Result<i32, Error> {
        check(unsafe { ffi::srack_patch_add_module(self.raw, t as c_int) })
    }
    pub fn set_field(&mut self, module: i32, field: i32, value: f64) -> Result<(), Error> {
        check(unsafe { ffi::srack_patch_set_field(self.raw, module, field, value) }).map(|_| ())
    }
    pub fn get_field(&self, module: i32, field: i32) -> Result<f64, Error> {
        let mut v = 0.0;
        check(unsafe { ffi::srack_patch_get_field(self.raw, module, field, &mut v) })?;
        Ok(v)
    }
    pub fn set_step(&mut self, module: i32, channel: i32, step: i32, state: i32, value: i32) -> Result<(), Error> {
        check(unsafe { ffi::srack_patch_set_step(self.raw, module, channel, step, state, value) }).map(|_| ())
    }
    /// What `WaveBox::load` leaves behind (src/synth/sample.rs:31-69): first channel as f32 + the file's sample rate.
    /// Edits between renders carry the modules' state over instead of restarting the voices (the reference's sliders do not reset state).
    pub fn keep_state(&mut self, keep: bool) -> Result<(), Error> {
        check(unsafe { ffi::srack_patch_keep_state(self.raw, keep as c_int) }).map(|_| ())
    }

    /// NoiseModule streams are keyed by (seed, module, first_voice + voice); the reference's generator is OS-seeded.
    pub fn set_noise_seed(&mut self, seed: u64, first_voice: u64) -> Result<(), Error> {
        check(unsafe { ffi::srack_patch_set_noise_seed(self.raw, seed, first_voice) }).map(|_| ())
    }

    pub fn set_wave(&mut self, module: i32, samples: &[f32], sample_rate: f32) -> Result<(), Error> {
        check(unsafe { ffi::srack_patch_set_wave(self.raw, module, samples.as_ptr(), samples.len() as u32, sample_rate) }).map(|_| ())
    }
    /// `SynthModuleWorkspaceImpl::deserialize` (src/ui.rs:116-135): a saved rack against this host's AudioConfig.
    pub fn load_srk(bytes: &[u8], sample_rate: u16, buffer_size: usize, channels: u8) -> Result<Self, Error> {
        let mut raw = std::ptr::null_mut();
        check(unsafe {
            ffi::srack_patch_load_srk(bytes.as_ptr() as *const c_void, bytes.len(), sample_rate as u32, buffer_size as u32, channels as u32, &mut raw)
        })?;
        Ok(Self { raw })
    }
    /// `serialize` (src/ui.rs:98-114).
    pub fn save_srk(&self) -> Result<Vec<u8>, Error> {
        let mut n = 0usize;
        check(unsafe { ffi::srack_patch_save_srk(self.raw, std::ptr::null_mut(), 0, &mut n) })?;
        let mut buf = vec![0u8; n];
        check(unsafe { ffi::srack_patch_save_srk(self.raw, buf.as_mut_ptr() as *mut c_void, n, &mut n) })?;
        Ok(buf)
    }
    /// `SynthModule::set_input(input_idx, src_module, src_port)`.
    pub fn set_input(&mut self, sink: i32, input_idx: u8, src: i32, src_port: u8) -> Result<(), Error> {
        check(unsafe { ffi::srack_patch_connect(self.raw, src, src_port as c_int, sink, input_idx as c_int) }).map(|_| ())
    }
    pub fn disconnect_input(&mut self, sink: i32, input_idx: u8) -> Result<(), Error> {
        check(unsafe { ffi::srack_patch_disconnect(self.raw, sink, input_idx as c_int) }).map(|_| ())
    }
    /// `plan_execution` as the workspace drives it: the execution order as module indices.
    pub fn plan(&mut self) -> Result<Vec<i32>, Error> {
        let mut order = vec![0 as c_int; 1024];
        let n = check(unsafe { ffi::srack_patch_plan(self.raw, order.as_mut_ptr(), order.len() as c_int) })?;
        order.truncate(n as usize);
        Ok(order)
    }
    pub fn configure_voices(&mut self, n_voices: u32) -> Result<(), Error> {
        check(unsafe { ffi::srack_voices_configure(self.raw, n_voices) }).map(|_| ())
    }
    pub fn set_voice_field(&mut self, module: i32, field: i32, values: &[f32]) -> Result<(), Error> {
        check(unsafe { ffi::srack_voices_set_field_f32(self.raw, module, field, values.as_ptr()) }).map(|_| ())
    }
    /// Number of distinct output planes and the plane of each channel (-1 = unconnected).
    pub fn planes(&mut self, channels: usize) -> Result<(usize, Vec<i32>), Error> {
        let mut cp = vec![0 as c_int; channels];
        let n = check(unsafe { ffi::srack_render_planes(self.raw, cp.as_mut_ptr(), channels as c_int) })?;
        Ok((n as usize, cp))
    }
    /// The batch counterpart of `execute(&plan)`: `n_samples` ticks of every voice, asynchronous on the default
    /// stream.  `frames`: `[planes][n_samples][n_voices]` f32, `mix`: `[channels][n_samples]` f32 (either may be None).
    pub fn execute_batch(&mut self, n_samples: u32, frames: Option<&DeviceBuffer>, mix: Option<&DeviceBuffer>, flags: u32) -> Result<(), Error> {
        let f = frames.map_or(std::ptr::null_mut(), |b| b.as_f32());
        let m = mix.map_or(std::ptr::null_mut(), |b| b.as_f32());
        check(unsafe { ffi::srack_render(self.raw, n_samples, f, m, flags, std::ptr::null_mut()) }).map(|_| ())
    }
}

impl Drop for Patch {
    fn drop(&mut self) {
        unsafe { ffi::srack_patch_destroy(self.raw) };
    }
}

pub fn abi_version() -> i32 {
    unsafe { ffi::srack_abi_version() }
}

/// The multi-GPU half: one process per GPU, voices sharded by global voice index, and ONE collective — the RCCL sum of the
/// per-rank partial mixes.  (No reference counterpart: s-rack is single-threaded, src/main.rs:59-63.)
/// Rank 0 makes the id; the host carries its 128 bytes to every rank; every rank calls `MixComm::new` with its device selected.
pub struct MixComm {
    raw: *mut c_void,
}

impl MixComm {
    pub fn unique_id() -> Result<[u8; DIST_ID_BYTES], Error> {
        let mut id = [0u8; DIST_ID_BYTES];
        check(unsafe { ffi::srack_dist_unique_id(id.as_mut_ptr()) })?;
        Ok(id)
    }
    pub fn new(id: &[u8; DIST_ID_BYTES], n_ranks: i32, rank: i32) -> Result<Self, Error> {
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::srack_dist_init(id.as_ptr(), n_ranks, rank, &mut raw) })?;
        Ok(MixComm { raw })
    }
    pub fn count(&self) -> Result<i32, Error> {
        let mut n = 0;
        check(unsafe { ffi::srack_dist_comm_count(self.raw, &mut n) })?;
        Ok(n)
    }
    /// ncclReduce(sum, f32) of the `[channels][n_samples]` partial mix, in place, on the default stream.
    pub fn reduce_mix(&self, mix: &DeviceBuffer, count: usize, root: i32) -> Result<(), Error> {
        check(unsafe { ffi::srack_dist_reduce_mix(self.raw, mix.as_f32(), count, root, std::ptr::null_mut()) }).map(|_| ())
    }
}

impl Drop for MixComm {
    fn drop(&mut self) {
        unsafe { ffi::srack_dist_destroy(self.raw) };
    }
}

#[cfg(test)]
mod tests {
    use super::*;

    /// oscillator::dco_tests::produces_440 (src/synth/oscillator.rs:284-305) through the GPU path.
    #[test]
    fn produces_440() {
        let mut p = Patch::new(440 * 4, 17, 2).unwrap();
        let osc = p.add_module(ModuleType::Oscillator).unwrap();
        let out = p.add_module(ModuleType::Output).unwrap();
        p.set_input(out, 0, osc, 0).unwrap();
        p.configure_voices(1).unwrap();
        let frames = DeviceBuffer::new(17 * 4).unwrap();
        let mut buf = [0f32; 17];
        p.execute_batch(17, Some(&frames), None, 0).unwrap();
        frames.to_host(&mut buf).unwrap();
        assert_eq!(buf[0], 0.0);
        assert!((buf[1] - 1.0).abs() < 0.00001);
        assert!(buf[2].abs() < 0.00001);
        assert!((buf[3] + 1.0).abs() < 0.00001);
        assert!(buf[4].abs() < 0.00001);
        p.execute_batch(17, Some(&frames), None, 0).unwrap();
        frames.to_host(&mut buf).unwrap();
        assert!((buf[0] - 1.0).abs() < 0.00001); // should continue smoothly into next buffer
    }
}
