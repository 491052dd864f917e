//! Safe wrappers over `ffi.rs` that keep the reference's names, argument meaning and error behaviour
//! (ssvgopal/nano-vllm-rs @ 2025-07-18; line numbers cite its files).  With `--features hip` these types replace
//! `src/engine/block_manager.rs`, `sequence.rs`, `scheduler.rs`, `model_runner.rs` and the body of
//! `LLMEngine::step`.  NOT compiled in the build image (no Rust toolchain); every call below names a function of
//! `ffi.rs`, which is generated from `include/nvr.h` and checked against the library's exports.
#![allow(dead_code)]
use crate::ffi;
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

pub fn last_error() -> String {
    unsafe { CStr::from_ptr(ffi::nvr_last_error()) }.to_string_lossy().into_owned()
}
/// `anyhow::Result` with the reference's message texts (block_manager.rs:159,163,225,267,280; scheduler.rs:219,236).
pub fn check(rc: c_int) -> anyhow::Result<()> {
    if rc == ffi::NVR_OK { Ok(()) } else { anyhow::bail!("{}", last_error()) }
}

// ---- Sequence (src/engine/sequence.rs:50-237): a handle; token_ids() / block_table() borrow from the library
pub struct Sequence { pub(crate) h: *mut ffi::nvr_seq_t, owned: bool }
impl Sequence {
    pub fn new(prompt_token_ids: Vec<i64>, sp: &SamplingParams, block_size: usize) -> Self {                  // :84
        let h = unsafe { ffi::nvr_seq_create(prompt_token_ids.as_ptr(), prompt_token_ids.len(), &sp.to_c(), block_size) };
        assert!(!h.is_null(), "{}", last_error());
        Self { h, owned: true }
    }
    pub fn seq_id(&self) -> u64 { unsafe { ffi::nvr_seq_id(self.h) } }
    pub fn len(&self) -> usize { unsafe { ffi::nvr_seq_len(self.h) } }                                        // :104
    pub fn num_completion_tokens(&self) -> usize { unsafe { ffi::nvr_seq_num_completion_tokens(self.h) } }    // :135
    pub fn num_cached_tokens(&self) -> usize { unsafe { ffi::nvr_seq_num_cached_tokens(self.h) } }
    pub fn num_blocks(&self) -> usize { unsafe { ffi::nvr_seq_num_blocks(self.h) } }                          // :157
    pub fn token_ids(&self) -> &[i64] {
        let (mut p, mut n) = (std::ptr::null(), 0usize);
        unsafe { ffi::nvr_seq_token_ids(self.h, &mut p, &mut n); std::slice::from_raw_parts(p, n) }
    }
    pub fn block_table(&self) -> &[i32] {
        let (mut p, mut n) = (std::ptr::null(), 0usize);
        unsafe { ffi::nvr_seq_block_table(self.h, &mut p, &mut n); std::slice::from_raw_parts(p, n) }
    }
    pub fn append_token(&mut self, t: i64) { unsafe { ffi::nvr_seq_append_token(self.h, t) } }                // :150
}
impl Drop for Sequence { fn drop(&mut self) { if self.owned { unsafe { ffi::nvr_seq_destroy(self.h) } } } }

// ---- SamplingParams (src/engine/sampling_params.rs:10-28)
#[derive(Clone, Debug)]
pub struct SamplingParams {
    pub temperature: f32, pub max_tokens: usize, pub ignore_eos: bool, pub top_p: Option<f32>, pub top_k: Option<usize>,
    pub repetition_penalty: Option<f32>,                                                                      // :27 (carried and validated, :112-117; the reference's sampler never reads it)
}
impl SamplingParams {
    pub fn to_c(&self) -> ffi::nvr_sampling_params {
        let mut c = unsafe { std::mem::zeroed::<ffi::nvr_sampling_params>() };
        unsafe { ffi::nvr_sampling_params_default(&mut c) };
        c.temperature = self.temperature; c.max_tokens = self.max_tokens as u64; c.ignore_eos = self.ignore_eos as i32;
        if let Some(p) = self.top_p { c.has_top_p = 1; c.top_p = p; }
        if let Some(k) = self.top_k { c.has_top_k = 1; c.top_k = k as u64; }
        if let Some(r) = self.repetition_penalty { c.has_repetition_penalty = 1; c.repetition_penalty = r; }
        c
    }
    pub fn validate(&self) -> anyhow::Result<()> { check(unsafe { ffi::nvr_sampling_params_validate(&self.to_c()) }) }   // :91-119
}

// ---- BlockManager (src/engine/block_manager.rs:69-361): same method names and Result types
pub struct BlockManager { h: *mut ffi::nvr_block_manager_t }
impl BlockManager {
    pub fn new(num_blocks: usize, block_size: usize) -> Self {                                                // :91 (asserts -> null)
        let h = unsafe { ffi::nvr_bm_create(num_blocks, block_size) };
        assert!(!h.is_null(), "{}", last_error());
        Self { h }
    }
    pub fn compute_hash(token_ids: &[i64], prefix_hash: Option<u64>) -> u64 {                                  // :109
        unsafe { ffi::nvr_bm_compute_hash(token_ids.as_ptr(), token_ids.len(), prefix_hash.is_some() as c_int, prefix_hash.unwrap_or(0)) }
    }
    pub fn can_allocate(&self, seq: &Sequence) -> bool { unsafe { ffi::nvr_bm_can_allocate(self.h, seq.h) != 0 } }                  // :152
    pub fn allocate(&mut self, seq: &mut Sequence) -> anyhow::Result<()> { check(unsafe { ffi::nvr_bm_allocate(self.h, seq.h) }) }  // :157
    pub fn deallocate(&mut self, seq: &mut Sequence) { let _ = unsafe { ffi::nvr_bm_deallocate(self.h, seq.h) }; }                  // :240
    pub fn can_append(&self, seq: &Sequence) -> bool { unsafe { ffi::nvr_bm_can_append(self.h, seq.h) != 0 } }                      // :255
    pub fn may_append(&mut self, seq: &mut Sequence) -> anyhow::Result<()> { check(unsafe { ffi::nvr_bm_may_append(self.h, seq.h) }) } // :265
    pub fn get_stats(&self) -> ffi::nvr_bm_stats {                                                                                  // :307
        let mut st = unsafe { std::mem::zeroed::<ffi::nvr_bm_stats>() };
        unsafe { ffi::nvr_bm_get_stats(self.h, &mut st) };
        st
    }
}
impl Drop for BlockManager { fn drop(&mut self) { unsafe { ffi::nvr_bm_destroy(self.h) } } }
unsafe impl Send for BlockManager {}        // one thread at a time, as behind the reference's Mutex (llm_engine.rs:25-28)

// ---- LLMEngine (src/engine/llm_engine.rs): step() collapses to one call; generate / generate_stream keep their signatures
pub struct LLMEngine { h: *mut ffi::nvr_engine_t }
pub struct SequenceOutput {                                                                                   // sequence.rs:30-47
    pub seq_id: u64, pub text: String, pub token_ids: Vec<i64>, pub completion_token_ids: Vec<i64>,
    pub num_prompt_tokens: usize, pub num_completion_tokens: usize, pub status: i32,
}
fn to_output(o: &ffi::nvr_sequence_output) -> SequenceOutput {               // copies out of the borrowed buffers
    let ids = unsafe { std::slice::from_raw_parts(o.token_ids, o.num_tokens) }.to_vec();
    let text = unsafe { std::slice::from_raw_parts(o.text as *const u8, o.text_len) };
    SequenceOutput { seq_id: o.seq_id, text: String::from_utf8_lossy(text).into_owned(),
                     completion_token_ids: ids[o.num_prompt_tokens..].to_vec(), token_ids: ids,
                     num_prompt_tokens: o.num_prompt_tokens, num_completion_tokens: o.num_completion_tokens, status: o.status }
}
impl LLMEngine {
    pub fn new(cfg: &ffi::nvr_config, mc: &ffi::nvr_model_config) -> anyhow::Result<Self> {                    // :34-63
        let h = unsafe { ffi::nvr_engine_create(cfg, mc) };
        if h.is_null() { anyhow::bail!("{}", last_error()) }
        Ok(Self { h })
    }
    pub fn add_request(&mut self, prompt: &[i64], sp: &SamplingParams) -> anyhow::Result<u64> {
        let mut id = 0u64;
        check(unsafe { ffi::nvr_engine_add_request(self.h, prompt.as_ptr(), prompt.len(), &sp.to_c(), &mut id) })?;
        Ok(id)
    }
    /// LLMEngine::step, :155-197: schedule -> execute_model -> sample_tokens -> postprocess inside the library.
    pub fn step(&mut self) -> anyhow::Result<ffi::nvr_step_info> {
        let mut info = unsafe { std::mem::zeroed::<ffi::nvr_step_info>() };
        check(unsafe { ffi::nvr_engine_step(self.h, &mut info) })?;
        Ok(info)
    }
    pub fn is_finished(&self) -> bool { unsafe { ffi::nvr_engine_is_finished(self.h) != 0 } }
    pub fn generate(&mut self, prompts: &[String], sp: &SamplingParams) -> anyhow::Result<Vec<SequenceOutput>> {   // :70-97
        let (ptrs, lens): (Vec<_>, Vec<_>) = prompts.iter().map(|p| (p.as_ptr() as *const c_char, p.len())).unzip();
        let (mut outs, mut n) = (std::ptr::null(), 0usize);
        check(unsafe { ffi::nvr_engine_generate(self.h, ptrs.as_ptr(), lens.as_ptr(), prompts.len(), &sp.to_c(), &mut outs, &mut n) })?;
        Ok(unsafe { std::slice::from_raw_parts(outs, n) }.iter().map(to_output).collect())
    }
    /// generate_stream, :100-128: the tokio channel stays on the Rust side; a closed receiver stops the stream (:250-253).
    pub fn generate_stream<F: FnMut(SequenceOutput) -> bool>(&mut self, prompts: &[String], sp: &SamplingParams, mut on_output: F) -> anyhow::Result<()> {
        unsafe extern "C" fn forward<F: FnMut(SequenceOutput) -> bool>(o: *const ffi::nvr_sequence_output, user: *mut c_void) -> c_int {
            let f = &mut *(user as *mut F);
            (!f(to_output(&*o))) as c_int                                    // false from the closure = receiver dropped
        }
        let (ptrs, lens): (Vec<_>, Vec<_>) = prompts.iter().map(|p| (p.as_ptr() as *const c_char, p.len())).unzip();
        check(unsafe { ffi::nvr_engine_generate_stream(self.h, ptrs.as_ptr(), lens.as_ptr(), prompts.len(), &sp.to_c(),
                                                       Some(forward::<F>), &mut on_output as *mut F as *mut c_void) })
    }
    pub fn get_stats(&self) -> ffi::nvr_engine_stats {                                                        // :312-327
        let mut st = unsafe { std::mem::zeroed::<ffi::nvr_engine_stats>() };
        unsafe { ffi::nvr_engine_get_stats(self.h, &mut st) };
        st
    }
    pub fn shutdown(&mut self) -> anyhow::Result<()> { check(unsafe { ffi::nvr_engine_shutdown(self.h) }) }   // :345-357
    /// Qwen3Model::load_weights (qwen3.rs:518-570) / ModelLoader (utils/loader.rs:43-198): one call per checkpoint tensor.
    pub fn load_tensor(&mut self, name: &str, dtype: c_int, shape: &[i64], data: &[u8]) -> anyhow::Result<()> {
        let cname = std::ffi::CString::new(name)?;
        let r = unsafe { ffi::nvr_engine_runner(self.h) };
        check(unsafe { ffi::nvr_runner_load_tensor(r, cname.as_ptr(), dtype, shape.as_ptr(), shape.len() as c_int, data.as_ptr() as *const c_void) })
    }
    /// Tensor parallel, one process per GPU: RCCL id from rank 0 + the ranks' hipIpc arena handles, gathered by the caller.
    pub fn init_tensor_parallel(&mut self, unique_id: &[u8; 128], all_handles: &[u8], devices: &[i32]) -> anyhow::Result<()> {
        let r = unsafe { ffi::nvr_engine_runner(self.h) };
        check(unsafe { ffi::nvr_runner_init_comm(r, unique_id.as_ptr()) })?;
        check(unsafe { ffi::nvr_runner_p2p_attach(r, all_handles.as_ptr(), devices.as_ptr()) })
    }
    pub fn export_p2p_handle(&mut self) -> anyhow::Result<[u8; 64]> {
        let mut h = [0u8; 64];
        check(unsafe { ffi::nvr_runner_p2p_export(ffi::nvr_engine_runner(self.h), h.as_mut_ptr()) })?;
        Ok(h)
    }
    /// After a step failed with NVR_ERR_RCCL (a peer never arrived inside a one-shot collective): EVERY rank calls this, then the
    /// control plane barriers, then the same engines take new requests (nvr.h, "recovery").
    pub fn recover_from_failed_collective(&mut self) -> anyhow::Result<()> {
        check(unsafe { ffi::nvr_engine_abort_last_batch(self.h) })?;
        check(unsafe { ffi::nvr_runner_p2p_reset(ffi::nvr_engine_runner(self.h)) })
    }
    /// The one-shot collectives' protocol (kernels/comm_p2p.hip): false = fence-free (default), true = r04's release / acquire fences.  Every rank alike, before
    /// the first step (the captured decode graphs are dropped); a control plane falls back fence-free -> fenced -> RCCL on a failed self-test.
    pub fn set_p2p_fenced(&mut self, on: bool) -> anyhow::Result<()> {
        check(unsafe { ffi::nvr_runner_p2p_set_fenced(ffi::nvr_engine_runner(self.h), on as i32) })
    }
    pub fn comm_selftest(&mut self) -> anyhow::Result<()> { check(unsafe { ffi::nvr_runner_comm_selftest(ffi::nvr_engine_runner(self.h)) }) }
}
impl Drop for LLMEngine { fn drop(&mut self) { unsafe { ffi::nvr_engine_destroy(self.h) } } }
unsafe impl Send for LLMEngine {}

// ---- Activation (src/layers/activation.rs:103-182): the enum, its FromStr and forward over device rows of the ops' 16-bit type
#[derive(Clone, Copy, Debug, PartialEq)]
pub enum ActivationType { SiLU = 0, GELU = 1, ReLU = 2, SiluAndMul = 3, GeluAndMul = 4 }
impl std::str::FromStr for ActivationType {                                                                  // :169-182 (the library owns the table: one spelling of it)
    type Err = anyhow::Error;
    fn from_str(s: &str) -> anyhow::Result<Self> {
        let c = std::ffi::CString::new(s)?;
        let mut t = -1i32;
        check(unsafe { ffi::nvr_activation_type_from_str(c.as_ptr(), &mut t) })?;
        Ok(match t { 0 => Self::SiLU, 1 => Self::GELU, 2 => Self::ReLU, 3 => Self::SiluAndMul, _ => Self::GeluAndMul })
    }
}
pub struct Activation { activation_type: ActivationType }
impl Activation {
    pub fn new(activation_type: ActivationType) -> Self { Self { activation_type } }                          // :121-145
    pub fn activation_type(&self) -> &ActivationType { &self.activation_type }                               // :162-164
    /// forward :147-159 on `rows` device rows of `cols` 16-bit elements; out has cols (SiLU / GELU / ReLU) or cols / 2 (the two fused types) columns
    pub unsafe fn forward(&self, x: *const u16, rows: i64, cols: i64, out: *mut u16, stream: *mut c_void) -> anyhow::Result<()> {
        check(ffi::nvr_activation(self.activation_type as i32, x, rows, cols, out, stream))
    }
}
