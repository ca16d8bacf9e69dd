/*
 * audiotoken_hip.h — C ABI of libaudiotoken_hip.so: the MI355X (gfx950) hot path of cmeraki/audiotoken.
 *
 * The reference has no FFI: its seam is the Python callable protocol
 *     self.encoder(input_batch: float32[B,N] on device, attention_mask: float32[B,N]) -> int16[B,K,T]
 * (reference audiotoken/core.py:194 and :276; encoder classes audiotoken/encoder.py:29-186) and
 *     self.decoder(tokens: int64[B,K,T]) -> float32[1, B*320*T]
 * (reference audiotoken/core.py:355-359; audiotoken/decoder.py:66-76).
 * Each entry point below names the reference interface it replaces. All pointers marked "device" are HIP
 * device pointers on the handle's device; everything is row-major and contiguous. Calls are stream-ordered and
 * never synchronise or allocate; scratch comes from a caller-provided workspace so the caller's allocator
 * (PyTorch's caching allocator in the Python binding) stays the only allocator. Functions return 0 on success,
 * a negative code on failure; at_last_error() returns the thread-local message. Nothing aborts.
 *
 * Threading: a handle is bound to one device; calls on one handle must be serialised by the caller; different
 * handles are independent (one process per GPU in the multi-GPU harness).
 */
#ifndef AUDIOTOKEN_HIP_H
#define AUDIOTOKEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* at_stream_t; /* hipStream_t */

/* ---- library ------------------------------------------------------------------------------------------ */
int at_version(void);
const char* at_last_error(void);

/* ---- acoustic tokenizer: EnCodec 24 kHz SEANet encoder + residual VQ, and the decoder ------------------
 * Replaces reference AcousticEncoder (audiotoken/encoder.py:29-57: ctor builds
 * EncodecModel.encodec_model_24khz(), forward = model.encoder -> model.quantizer.encode -> transpose ->
 * int16) and AcousticDecoder (audiotoken/decoder.py:50-76). */
typedef struct at_encodec at_encodec_t;

/* Create an empty model bound to `device_id` (replaces encoder.py:38-39 `EncodecModel...to(device)`). */
at_encodec_t* at_encodec_create(int device_id);

/* Hand one named host tensor (float32) to the model. Names are the `encodec` checkpoint keys with the
 * weight-norm pair already folded to a plain `.weight` (W = g*v/||v||, the tensor the reference convolves
 * with): "encoder.model.0.conv.conv.weight" [32,1,7] / ".bias", "encoder.model.{1,4,7,10}.block.{1,3}.conv.conv.*",
 * "encoder.model.{1,4,7,10}.shortcut.conv.conv.*", "encoder.model.{3,6,9,12}.conv.conv.*",
 * "encoder.model.13.lstm.{weight_ih,weight_hh,bias_ih,bias_hh}_l{0,1}", "encoder.model.15.conv.conv.*",
 * "quantizer.vq.layers.{k}._codebook.embed" [1024,128] (optional ".e2" [1024] = row-wise sum of squares),
 * and for decode "decoder.model.*" ("…convtr.convtr.weight" is [in,out,k]). */
int at_encodec_set_tensor(at_encodec_t* h, const char* name, const float* host_data, const int64_t* shape, int ndim);

/* Repack to kernel layouts, upload, free host staging. with_decoder != 0 also requires the decoder tensors. */
int at_encodec_finalize(at_encodec_t* h, int with_decoder);
void at_encodec_destroy(at_encodec_t* h);
int at_encodec_num_codebooks(const at_encodec_t* h);

/* Workspace size for at_encodec_encode on B clips of N samples. */
size_t at_encodec_workspace_bytes(const at_encodec_t* h, int B, int N);

/* Replaces AcousticEncoder.forward (audiotoken/encoder.py:44-57).
 *   wav   device float32 [B][N]           (24 kHz mono, any N >= 1 with N > 9 so reflect padding is defined)
 *   mask  device float32 [B][N] or NULL   — ignored, exactly as the reference ignores attention_mask
 *   n_q   number of codebooks in {1..loaded}; the reference derives it from the bandwidth (encoder.py:50-52)
 *   codes device int16   [B][n_q][T], T = ceil(N/320) returned through *T_out (may be NULL)
 *   emb_out optional device float32 [B][T][128]: the pre-quantiser embedding (parity taps; NULL in production) */
int at_encodec_encode(at_encodec_t* h, const float* wav, const float* mask, int B, int N, int n_q, int16_t* codes,
                      int* T_out, float* emb_out, void* workspace, size_t workspace_bytes, at_stream_t stream);

/* Same as at_encodec_encode, plus a device uint32 status word (stream-ordered): 0 on success; bit 0 (1) = a bounded wait inside the
 * persistent LSTM kernel gave up (the call still terminates); bit 1 (2) = an activation did not fit the fp16 range of an f16x2 kernel ("chain_f16x2",
 * "ih_f16x2", "res_f16x2", "rvq_f16x2", "fin_f16x2"). In both cases the codes are invalid and the caller repeats the batch on the safe path (options
 * persistent_lstm = 0 / chain_f16x2 = ih_f16x2 = res_f16x2 = rvq_f16x2 = fin_f16x2 = 0). Bit 2 (4) = a NaN or an infinity reached the RVQ search (a
 * non-finite sample in `wav`, as a rule): repeating does not help; the reference emits arbitrary codes for such input without a diagnostic.
 * THREADING: a handle carries per-call bookkeeping (the range table the status word is combined from, zeroed at the start of every call), so the
 * *_checked / range_report calls of ONE handle must be issued on one stream at a time — two concurrent calls on the same handle (an encode and a
 * decode included) can clear each other's flags. Use one handle per stream. The same holds for at_w2vbert_* and at_hubert_* handles and for the
 * handle-less at_op_*_split entry points (one range pair per device). */
int at_encodec_encode_checked(at_encodec_t* h, const float* wav, const float* mask, int B, int N, int n_q, int16_t* codes,
                              int* T_out, float* emb_out, void* workspace, size_t workspace_bytes, at_stream_t stream,
                              uint32_t* status_dev);

/* Options (they select kernels or bound memory; all but the "*_x3" / "*_f16x2" ones leave the results bit-identical). The product runs the
 * defaults; the others exist as the per-batch range fallback of AcousticEncoder.verified ("*_f16x2" = 0: three bf16 pieces, fp32 exponent range), as
 * the machine fallback ("persistent_lstm" = 0) and as the A/B twins the parity tests compare against (fp32 kernels, unfused GEMM paths). Environment
 * switches are limited to $AUDIOTOKEN_BF16X3_ACOUSTIC, $AUDIOTOKEN_X3_KERNELS (A/B masks of tools/ab_x3.sh), $AUDIOTOKEN_LSTM_STEPWISE and
 * $AUDIOTOKEN_SUBBATCH: round 2's per-option variables were removed.
 *   "persistent_lstm" 1/0 — whole-sequence persistent LSTM kernel (default on) or one launch per time step;
 *   "fused_stage0", "fused_res64", "fused_res128", "fused_down64", "fused_dectail" 1/0 — fused SEANet kernels (default on)
 *   or the GEMM path; "fused_stage1" 1/0 — the 64-channel block and the stage-1 strided conv in ONE kernel (seanet_res64down.hip, default on;
 *   needs "fused_res64", "fused_down64", "res64_x3", "down64_x3", "res_f16x2"; bit-identical to the two kernels it replaces);
 *   "stage0_x3", "res64_x3", "res128_x3", "down64_x3", "down128_x3", "down256_x3", "res256_x3", "lstm_x3", "rvq_x3" 1/0 — the fused kernels, the
 *   stage-2 / stage-3 convs + 256-channel block (as chained GEMMs) the LSTM recurrence and the RVQ search on the bf16 matrix cores with exact 3-way bf16 splits of every
 *   operand (default on; $AUDIOTOKEN_X3_KERNELS bit mask, bits 3, 2, 1, 0, 4, 5, 6, 7, 8 in that order; 0 = the fp32-MFMA kernels:
 *   same tokens, embeddings differ in the last bits);
 *   "ih_f16x2", "chain_f16x2" 1/0 — the two LSTM input projections (encoder and decoder) / the encoder's stage-2 strided conv, 256-channel block
 *   and stage-3 strided conv (the GEMM chain) as two-piece fp16 operand splits, three MFMA products (default on)
 *   or as the three-piece bf16 splits, six products; see csrc/gemm_bf16x3.h;
 *   "res_f16x2", "rvq_f16x2" 1/0 — the fused SEANet kernels of the encoder (stage 0, the 64- and 128-channel residual blocks, the stage-1 strided conv) /
 *   the RVQ search on the same two-piece fp16 scheme (default on) or on three bf16 pieces;
 *   "res128_rs" 1/0 — the 128-channel block on the fp16 scheme as the role-split kernel (csrc/seanet_res128rs.hip, default) or as
 *   csrc/seanet_res128x3.hip; bit-identical results;
 *   "up_f16x2" 1/0 — the decoder's first three transposed convs as two-tap windowed split GEMMs on the fp16 scheme (default on) or as fp32-MFMA GEMMs;
 *   "fin_f16x2" 1/0 — the encoder's final k = 7 conv as a windowed split GEMM on the two-piece fp16 scheme (default on) or on
 *   the fp32 MFMA;
 *   "lstm_f16x2" 1/0 — the persistent LSTM's recurrent product on the two-piece fp16 scheme (h in (-1, 1) always fits: no range check) or on
 *   three bf16 pieces (default on; needs "lstm_x3" = 1);
 *   "lstm_spin_limit" n >= 0 — polls of a hand-off flag before a workgroup of the persistent LSTM gives up and the status word of the
 *   *_checked entry points becomes 1 (default 2^18, i.e. 0.1-0.3 s; 0 makes the first unready poll give up — used by the tests);
 *   "subbatch" n >= 1 — clips per pass through the conv stack (default 256 or $AUDIOTOKEN_SUBBATCH): bounds
 *   at_encodec_workspace_bytes / at_encodec_decode_workspace_bytes, which must be re-queried after changing it. */
int at_encodec_set_option(at_encodec_t* h, const char* name, int value);
/* Current value of an option of at_encodec_set_option, or -1 for an unknown name. */
int at_encodec_get_option(const at_encodec_t* h, const char* name);

/* Optional timing taps for the benchmark: when enabled, encode brackets each kernel group (conv0, res0..3,
 * down0..3, lstm_ih, lstm_rec, final_conv, rvq) with HIP events recorded on the launch stream.
 * at_encodec_profile(h, enable) resets the accumulated spans. at_encodec_profile_read synchronises on the
 * recorded events and returns the number of groups (names '\n'-separated), or a negative error code. */
int at_encodec_profile(at_encodec_t* h, int enable);
int at_encodec_profile_read(at_encodec_t* h, char* names, size_t names_cap, float* total_ms, int* launches, int max_groups);

size_t at_encodec_decode_workspace_bytes(const at_encodec_t* h, int B, int T);

/* Replaces AcousticDecoder.forward (audiotoken/decoder.py:66-76): codes device int64 [B][K][T] ->
 * wav device float32 [B*320*T] (the reference's [1, B*320*T] row). */
int at_encodec_decode(at_encodec_t* h, const int64_t* codes, int B, int K, int T, float* wav, void* workspace,
                      size_t workspace_bytes, at_stream_t stream);
/* Same, plus the device status word of at_encodec_encode_checked (the decoder runs the same persistent LSTM). */
int at_encodec_decode_checked(at_encodec_t* h, const int64_t* codes, int B, int K, int T, float* wav, void* workspace,
                              size_t workspace_bytes, at_stream_t stream, uint32_t* status_dev);

/* ---- semantic_m tokenizer: log-mel front-end + Wav2Vec2-BERT conformer + LayerNorm + VQ ------------------
 * Replaces reference Wav2VecBertEncoder (audiotoken/encoder.py:111-186): ctor = Wav2VecBertProcessor +
 * Wav2Vec2BertModel.from_pretrained + VectorQuantize(dim=1024, codebook_size=2048) (:112-161); forward =
 * processor -> model(..., output_hidden_states=True).hidden_states[output_layer] -> LayerNorm(no affine) ->
 * vq -> int16 [B,1,T'] (:163-184), with the attention of audiotoken/modeling_wav2vec2_bert.py:20-80. */
typedef struct at_w2vbert at_w2vbert_t;

at_w2vbert_t* at_w2vbert_create(int device_id);

/* Named host tensors (float32): the HF Wav2Vec2BertModel state-dict keys ("feature_projection.*",
 * "encoder.layers.{i}.*" for consecutive i from 0), "vq._codebook.embed" [1,2048,1024] (the VectorQuantize
 * state-dict key, reference audiotoken/utils.py:331-339; optional "vq._codebook.e2" [2048]), and the two front-end
 * tables of reference processors.py:66-78: "frontend.window" [400], "frontend.mel_filters" [257,80]. */
int at_w2vbert_set_tensor(at_w2vbert_t* h, const char* name, const float* host_data, const int64_t* shape, int ndim);
int at_w2vbert_finalize(at_w2vbert_t* h);
/* The finalized model as ONE device blob, for start-up at N > 1 (weights cross xGMI once): one rank reads, folds, uploads and splits
 * the checkpoint and exports; the others import what one RCCL broadcast delivered — no D2H copy, no second host pass over 1.8 GB, no split kernels.
 * (The reference has no counterpart: audiotoken/core.py:66 is single-device; every process would call from_pretrained itself.)
 *   packed_bytes(h)                       size of the blob of a finalized handle (-1: not finalized)
 *   packed_meta(h, host_dst, cap)         the host-side record (block sizes, max |w| per tensor, layer count, arithmetic): returns its size, writes it
 *                                         when cap is large enough (call with NULL / 0 first)
 *   export_packed(h, device_dst, bytes, stream)   concatenate the handle's device allocations into device_dst (stream-ordered D2D copies)
 *   import_packed(h, meta, meta_bytes, device_src, bytes, stream)   on a FRESH handle (create only): copy the blob into one allocation owned by the handle
 *                                         and rebuild the model over it; the handle is finalized afterwards, device_src may be freed. Fails (-1,
 *                                         at_last_error) when the record does not match this build's allocation order / sizes.
 * Both sides must be the same build of this library on the same architecture; the blob is not a file format. */
int64_t at_w2vbert_packed_bytes(at_w2vbert_t* h);
int64_t at_w2vbert_packed_meta(at_w2vbert_t* h, void* host_dst, int64_t cap);
int at_w2vbert_export_packed(at_w2vbert_t* h, void* device_dst, int64_t bytes, void* stream);
int at_w2vbert_import_packed(at_w2vbert_t* h, const void* host_meta, int64_t meta_bytes, const void* device_src, int64_t bytes, void* stream);
void at_w2vbert_destroy(at_w2vbert_t* h);
int at_w2vbert_num_layers(const at_w2vbert_t* h);

/* T' = pad_to_multiple(floor((1 + floor((N-400)/160)) / 2)) (reference processors.py:158,246-259). */
int at_w2vbert_num_tokens(int N, int pad_to_multiple_of);
size_t at_w2vbert_workspace_bytes(const at_w2vbert_t* h, int B, int N, int pad_to_multiple_of);

/* Replaces Wav2VecBertEncoder.forward (audiotoken/encoder.py:163-184).
 *   wav device float32 [B][N] (16 kHz); mask device float32 [B][N] (1 = real sample) or NULL (= all ones)
 *   pad_to_multiple_of: the reference's third argument (default 2)
 *   n_layers: conformer layers to run = the reference's output_layer (19; hidden_states[n] is the output of layer n-1)
 *   tokens device int16 [B][1][T'] or NULL (then no VQ); *T_out = T'
 *   parity taps, all optional (NULL in production): features_out [B][T'][160], attn_mask_out [B][T'],
 *   hidden_out [B][T'][1024] = hidden_states[n_layers]. */
int at_w2vbert_encode(at_w2vbert_t* h, const float* wav, const float* mask, int B, int N, int pad_to_multiple_of, int n_layers,
                      int16_t* tokens, int* T_out, float* features_out, float* attn_mask_out, float* hidden_out, void* workspace,
                      size_t workspace_bytes, at_stream_t stream);
/* Same as at_w2vbert_encode, plus a device int32 status word (stream-ordered: zeroed at the start of the call, final when the call's work
 * has completed): bit 1 (value 2) = an activation did not fit the fp16 range of the "f16x2" arithmetic (|x| > 65504 / 16) — the
 * tokens are then invalid and the caller should repeat the batch after at_w2vbert_set_option(h, "arith", 1); bit 2 (value 4) = a NaN or an
 * infinity reached the quantiser (a non-finite sample in `wav`): repeating does not help. One stream per handle (see at_encodec_encode_checked). */
int at_w2vbert_encode_checked(at_w2vbert_t* h, const float* wav, const float* mask, int B, int N, int pad_to_multiple_of, int n_layers,
                              int16_t* tokens, int* T_out, float* features_out, float* attn_mask_out, float* hidden_out, void* workspace,
                              size_t workspace_bytes, at_stream_t stream, int32_t* status_dev);
/* Options. "arith": arithmetic of the eight linear layers per conformer layer — 0 = f32-input MFMA; 1 = "bf16x3": exact 3-way bf16
 * operand splits, six products on the bf16 matrix cores; 2 = "f16x2" (default, also $AUDIOTOKEN_SEMANTIC_ARITH=f32|bf16x3|f16x2): two fp16
 * pieces per operand, three products, operands pre-scaled by powers of two (fp32-class accuracy: csrc/gemm_bf16x3.h). Weights are split
 * for a scheme the first time it is selected. "dwconv_stream" 1/0 (default 1): the conv module's depthwise conv + LayerNorm + swish as the streaming
 * kernel (csrc/dwconv_stream.hip: one channel per thread walking along time) or the register-stationary one — bit-identical results, the option is
 * the A/B twin the tests compare. "vq_split" 1/0 (default 1): the VQ score GEMM (LayerNorm output x code book) on the split kernel with the handle's
 * arithmetic, or on the fp32 MFMA (always with "arith" = 0). at_w2vbert_get_option returns the current value (or -1). */
int at_w2vbert_set_option(at_w2vbert_t* h, const char* name, int value);
int at_w2vbert_get_option(const at_w2vbert_t* h, const char* name);
int at_w2vbert_profile(at_w2vbert_t* h, int enable);
int at_w2vbert_profile_read(at_w2vbert_t* h, char* names, size_t names_cap, float* total_ms, int* launches, int max_groups);

/* ---- semantic_s tokenizer: mHuBERT-base + k-means ---------------------------------------------------------
 * Replaces reference HubertEncoder (audiotoken/encoder.py:60-108): ctor = HubertModel.from_pretrained + joblib k-means
 * centres (:61-85); __call__ = model(..., output_hidden_states=True).hidden_states[output_layer] -> LayerNorm(no
 * affine) -> torch.cdist -> argmin -> int16 [B,1,T] (:87-108). Input is the waveform AFTER hubert_processor
 * (zero-mean / unit-variance, encoder.py:20-26), which the reference applies on the host before batching. */
typedef struct at_hubert at_hubert_t;
at_hubert_t* at_hubert_create(int device_id);
/* HF HubertModel state-dict keys; the positional conv's weight-norm pair folded by the caller (dim = 2) into
 * "encoder.pos_conv_embed.conv.weight" [768,48,128]; "kmeans.cluster_centers_" [1000,768] (optional "kmeans.c2" [1000]). */
int at_hubert_set_tensor(at_hubert_t* h, const char* name, const float* host_data, const int64_t* shape, int ndim);
int at_hubert_finalize(at_hubert_t* h);
/* the finalized model as one device blob: as at_w2vbert_packed_bytes / _packed_meta / _export_packed / _import_packed above */
int64_t at_hubert_packed_bytes(at_hubert_t* h);
int64_t at_hubert_packed_meta(at_hubert_t* h, void* host_dst, int64_t cap);
int at_hubert_export_packed(at_hubert_t* h, void* device_dst, int64_t bytes, void* stream);
int at_hubert_import_packed(at_hubert_t* h, const void* host_meta, int64_t meta_bytes, const void* device_src, int64_t bytes, void* stream);
void at_hubert_destroy(at_hubert_t* h);
int at_hubert_num_layers(const at_hubert_t* h);
/* T = chained floor((L - k)/s) + 1 over the 7 feature-extractor convs (HF modeling_hubert.py:664-677). */
int at_hubert_num_tokens(int N);
size_t at_hubert_workspace_bytes(const at_hubert_t* h, int B, int N);
/* wav device float32 [B][N] (normalised, 16 kHz); mask device float32 [B][N] or NULL; n_layers = output_layer (11);
 * tokens device int16 [B][1][T] or NULL; hidden_out optional device float32 [B][T][768] = hidden_states[n_layers]. */
int at_hubert_encode(at_hubert_t* h, const float* wav, const float* mask, int B, int N, int n_layers, int16_t* tokens, int* T_out,
                     float* hidden_out, void* workspace, size_t workspace_bytes, at_stream_t stream);
/* As at_w2vbert_encode_checked / at_w2vbert_set_option / at_w2vbert_get_option: device status word (bit 1 = fp16 range overflow of the
 * "f16x2" arithmetic) and the "arith" option (0 f32 MFMA, 1 bf16x3, 2 f16x2 = default; also covers the six 512->512 feature-extractor convs). */
int at_hubert_encode_checked(at_hubert_t* h, const float* wav, const float* mask, int B, int N, int n_layers, int16_t* tokens, int* T_out,
                             float* hidden_out, void* workspace, size_t workspace_bytes, at_stream_t stream, int32_t* status_dev);
int at_hubert_set_option(at_hubert_t* h, const char* name, int value);
int at_hubert_get_option(const at_hubert_t* h, const char* name);
int at_hubert_profile(at_hubert_t* h, int enable);
int at_hubert_profile_read(at_hubert_t* h, char* names, size_t names_cap, float* total_ms, int* launches, int max_groups);

/* ---- input side of encode_batch_files: what the reference does through torchaudio / ffmpeg ------------------------------- */
/* FLAC (RFC 9639), host code: replaces the StreamReader decode of reference audiotoken/utils.py:71-101 for '.flac' members of AUDIO_EXTS. `data` = the whole
 * file in host memory. at_flac_info: STREAMINFO fields (md5_16 nullable: the MD5 of the decoded little-endian PCM, for the caller to verify).
 * at_flac_decode: planar int32 samples out[c * cap_samples_per_channel + i], every frame's CRC-8 / CRC-16 verified; returns samples per channel or a
 * negative code. Thread-safe (no handle, no device). */
int at_flac_info(const uint8_t* data, size_t n, int* sample_rate, int* channels, int* bits_per_sample, int64_t* total_samples, uint8_t* md5_16);
int64_t at_flac_decode(const uint8_t* data, size_t n, int32_t* out, int64_t cap_samples_per_channel);

/* Raw PCM on the device -> the batch the encoders take, in one launch: sample format conversion, the per-chunk resampling of reference
 * audiotoken/utils.py:82-98 (torchaudio Resample defaults), the segmentation / zero padding / mask of reference audiotoken/datasets.py:75-105.
 * One descriptor per output row (built on the host from headers and lengths only; the array lives in device memory):
 *   pcm        device pointer to the decoded file's mono samples in format `fmt`
 *   table      device pointer to the resampling table of (orig, new): float32 [n][2 width + o] (audiotoken_amd/audio_io.py: resample_table) followed by
 *              int32 [n][2] = the non-zero tap range [lo, hi) of every phase; NULL = the file is at the model's rate
 *   chunk_off / chunk_len   the streamed chunk (chunk_size seconds at the SOURCE rate) this row is cut from: every chunk is resampled on its own
 *   out_start / valid_len   the row = samples [out_start, out_start + valid_len) of the resampled chunk, then padding up to seg_len
 *   scale      multiplies integer samples (1 / 32768 for 16-bit WAV, 1 / 2^31 for 24 / 32-bit WAV, 1 / 2^(bits - 1) for FLAC); u8: (x - 128) * scale
 *   o, n, width   orig / g, new / g, half kernel width (o == n when table is NULL)
 * segments [nseg][seg_len] float32, masks [nseg][seg_len] float32 (1 = sample exists; nullable). Stream-ordered, no allocation. */
enum { AT_PCM_S16 = 0, AT_PCM_S32 = 1, AT_PCM_F32 = 2, AT_PCM_U8 = 3 };
typedef struct at_segment_desc {
    const void* pcm;
    const float* table;
    int64_t chunk_off;
    int32_t chunk_len, out_start, valid_len, fmt;
    float scale;
    int32_t o, n, width;
    int32_t chunk_out_len;   /* samples of the whole resampled chunk (ceil(n * chunk_len / o); chunk_len at the model's rate): the span of the per-chunk moments below */
} at_segment_desc;
int at_segments_from_pcm(const at_segment_desc* descs_dev, int nseg, int seg_len, float pad_value, float* segments, float* masks, at_stream_t stream);
/* The same with the reference's per-chunk transform of Tokenizers.semantic_s folded in (round 5): `hubert_processor` = HF Wav2Vec2FeatureExtractor with
 * do_normalize (reference audiotoken/encoder.py:20-26), applied by the reference to every streamed chunk BEFORE it is cut and padded
 * (audiotoken/datasets.py:78-79): valid samples become (x - mean) / sqrt(var + eps) with mean / population variance over the row's whole resampled chunk
 * (float64 sums in a fixed order: deterministic), eps = 1e-7; padding stays pad_value. workspace: at_segments_zmuv_workspace_bytes(nseg, max over rows of
 * chunk_out_len) bytes of device memory. Three launches, stream-ordered, no allocation. */
size_t at_segments_zmuv_workspace_bytes(int nseg, int max_chunk_out_len);
int at_segments_from_pcm_zmuv(const at_segment_desc* descs_dev, int nseg, int seg_len, int max_chunk_out_len, float pad_value, float eps, float* segments,
                              float* masks, void* workspace, size_t workspace_bytes, at_stream_t stream);

/* ---- measurement aid (bench.py): the clock the chip held over a stretch of the stream ----------------------------------------------------------------
 * slots_dev: device uint64 [16][2], zeroed by the caller. One tiny launch writes {s_memtime (shader cycles), s_memrealtime (100 MHz)} into slot [xcc] for every
 * XCD a wave of it ran on. Between two stamps A, B on one stream: held clock = (B.memtime - A.memtime) / (B.memrealtime - A.memrealtime) x 100 MHz per XCD
 * (XCDs' counters are not mutually synchronised). No product kernel carries a stamp; the reference has no counterpart. */
int at_clock_stamp(uint64_t* slots_dev, at_stream_t stream);

/* ---- operator-level entry points (the kernels behind the models; used by the parity tests) ------------- */

/* Measured fp16 headroom of the LAST encode / decode of a handle (round 3). Every place where activations are split into fp16 pieces (a "site":
 * a fused SEANet kernel's staging, a GEMM epilogue that writes the next layer's operand, LayerNorm -> pieces, the attention kernel ...) records the
 * largest |x * scale| it saw; the two-piece fp16 arithmetic overflows at 65504 (status bit 1). at_*_range_sites: newline-separated site names
 * (returns the count, or -(bytes needed)); at_*_range_report: max_scaled[k] per site (0 = the site did not run on the fp16 scheme), returns the
 * count. Synchronises the device. */
int at_encodec_range_sites(char* names, size_t cap);
int at_encodec_range_report(at_encodec_t* h, float* max_scaled, int cap);
int at_w2vbert_range_sites(char* names, size_t cap);
int at_w2vbert_range_report(at_w2vbert_t* h, float* max_scaled, int cap);
int at_hubert_range_sites(char* names, size_t cap);
int at_hubert_range_report(at_hubert_t* h, float* max_scaled, int cap);
/* The power of two each split site multiplies its activations by before the fp16 split (round 5). 16 everywhere, except that a site fed by an affine
 * LayerNorm gets, at finalize, the largest power of two s <= 16 with s (sqrt(D) max|gamma| + max|beta|) <= 65 000: |LN(x)_k| <= sqrt(D) |gamma_k| + |beta_k|
 * for any input, so those sites provably cannot overflow whatever the data. at_w2vbert_site_scales: scales[l * n_sites + k] (n_sites and the order of
 * at_w2vbert_range_sites); at_hubert_site_scales: scales[2 l] = input of layer l's q/k/v projection, scales[2 l + 1] = input of its first FFN GEMM. Return
 * the number of floats written. (The reference has no such notion: fp32 throughout, audiotoken/encoder.py:87-108,163-186.) */
int at_w2vbert_site_scales(const at_w2vbert_t* h, float* scales, int cap);
int at_hubert_site_scales(const at_hubert_t* h, float* scales, int cap);
/* Per conformer layer, the OR of its split sites' status flags in the LAST encode (bit 1 = an activation of that layer left the fp16 range). An overflow
 * becomes infinities that every later layer flags as well: the FIRST flagged layer is the cause. Returns the number of layers written (<= cap).
 * Synchronises the device. With option "layer_arith:<i>" (at_w2vbert_set_option; -1 = the handle's "arith", 1 = bf16x3, 2 = f16x2) ONE layer can be moved
 * to the wide-range arithmetic while the others stay on f16x2 — what the product's range fallback does (round 4; the reference has no such notion:
 * its fp32 arithmetic cannot overflow, audiotoken/encoder.py:163-186). */
int at_w2vbert_layer_status(at_w2vbert_t* h, int32_t* flags, int cap);
/* The same for at_hubert_t: flags[0] = conv feature encoder + positional conv, flags[1 + l] = transformer layer l; option "layer_arith:<l>" of
 * at_hubert_set_option moves one transformer layer to bf16x3 / f16x2. */
int at_hubert_layer_status(at_hubert_t* h, int32_t* flags, int cap);

/* Windowed fp32 GEMM: out[b][m][n] = act(alpha*(sum_kk A(b,m,kk)*W[n][kk] + bias[n])) (+ R[b][m][n]) with
 * A(b,m,kk) = pro(X[b][m*stride + kk/Cin - pad_left][kk%Cin]); rows outside [0,Tin) reflect (pad_mode=1) or are
 * zero (pad_mode=0). Covers conv1d (ref: encodec SConv1d), Linear and ConvTranspose1d-as-phases. */
typedef struct at_gemm_desc {
    const float* X; int64_t x_bstride; int32_t Tin, Cin, ldx;
    int32_t ktaps, stride, pad_left, pad_mode;
    const float* W; const float* bias;
    float* C; int64_t c_bstride; int32_t ldc;
    const float* R; int64_t r_bstride; int32_t ldr;
    int32_t M, N, K, batch;
    int32_t pro; /* 0 none, 1 ELU, 2 power: A = X[kk]^2 + X[kk+aux_off]^2 */
    int32_t epi; /* 0 none, 1 swish, 2 ELU, 3 GELU(erf), 4 log(max(v, mel_floor)), 5 GLU over interleaved rows */
    float alpha;
    int32_t aux_off;
    const float* row_mask; /* optional [batch*M]: rows with mask 0 are written as zeros */
} at_gemm_desc;
int at_op_gemm(const at_gemm_desc* d, at_stream_t stream);

/* Residual VQ search (ref: encodec ResidualVectorQuantizer.encode; formula SURVEY.md Appendix A.1):
 * x device float32 [rows][128]; codebooks device [n_q][1024][128]; e2 device [n_q][1024];
 * codes int16 written at codes[(row / T)*n_q*T + q*T + row % T]. */
/* The named host tensors (float32, weight-norm already folded) a model's finalize() needs, one per line as "name d0 d1 ...":
 * model "encodec" (n = number of codebooks, with_extras = decoder too), "w2vbert" (n = conformer layers, with_extras = VQ codebook),
 * "hubert" (n = transformer layers, with_extras = k-means centres). Returns the number of entries, or -(bytes needed) when buf is NULL or
 * too small. A checkpoint loader can validate its output against this list without a device. */
int at_required_tensors(const char* model, int n, int with_extras, char* buf, size_t cap);

/* The split-operand GEMM of the semantic tokenizers (csrc/gemm_bf16x3.h), for the parity tests: C[M][N] = X[M][K] . W[N][K]^T + bias with both
 * fp32 operands written as 16-bit pieces — scheme 0 = three bf16 pieces / six products, 1 = two fp16 pieces / three products (w_max_abs = max |W|
 * sets the weight scale). kernel: 0 = as the product dispatches, 1 = the two-group kernel (gemm_f16x2_tg.hip), 2 = the register-staged kernel.
 * workspace >= (round_up(M, 256) + N) * K * pieces * 2 bytes. status_dev: nullable device word (bit 1 = fp16 range overflow). */
int at_op_gemm_split(const float* X, const float* W, const float* bias, float* C, int M, int N, int K, int scheme, float w_max_abs,
                     int kernel, void* workspace, size_t workspace_bytes, int32_t* status_dev, at_stream_t stream);
/* A causal conv1d (reflect front padding k - stride; ref: encodec SConv1d) on the WINDOWED two-piece fp16 split GEMM
 * (csrc/gemm_f16x2_tg.hip), for the parity / soak tests: X [B][L][Cin] float32 channels-last, W [Cout][k * Cin] (tap-major), C [B][L / stride][Cout].
 * Cin % 16, Cout % 128, L % stride == 0. workspace >= 2 * (B * Cin * stride * (round_up(L / stride, 256) + (k - 1) / stride + 1) + Cout * k * Cin) * 2 bytes. */
int at_op_conv_split(const float* X, const float* W, const float* bias, float* C, int B, int L, int Cin, int Cout, int ktaps, int stride,
                     float w_max_abs, void* workspace, size_t workspace_bytes, int32_t* status_dev, at_stream_t stream);
int at_op_rvq_encode(const float* x, int64_t rows, int T, const float* codebooks, const float* e2, int n_q,
                     int16_t* codes, at_stream_t stream);
/* The same search on the SPLIT kernel the product runs (csrc/rvq_encode_x3.hip): scheme 1 = two fp16 pieces / three products (default), 0 = three
 * bf16 pieces / six products; cb_max_abs = max |codebook| (the fp16 scheme's power-of-two codebook scale); workspace >= pieces * n_q * 1024 * 128 * 2 + 8
 * bytes; status_dev nullable (bit 1 = fp16 range overflow of the residual). */
int at_op_rvq_encode_split(const float* x, int64_t rows, int T, const float* codebooks, const float* e2, int n_q, int16_t* codes, int scheme,
                           float cb_max_abs, void* workspace, size_t workspace_bytes, int32_t* status_dev, at_stream_t stream);

/* LayerNorm over the last dim (eps 1e-5); gamma/beta NULL = non-affine; rows with row_mask 0 are zeroed. */
int at_op_layernorm(const float* x, const float* gamma, const float* beta, const float* row_mask, float* y, int64_t rows, int D,
                    at_stream_t stream);

/* Rel-pos attention (ref audiotoken/modeling_wav2vec2_bert.py:46-73): qkv [B*T][3072] = [q|k|v] (16 heads x 64),
 * attn_mask [B*T] (1 = valid key), dist_emb80 [80][64] (73 rows used, rest zero) -> ctx [B*T][1024]. */
int at_op_relpos_attention(const float* qkv, const float* attn_mask, const float* dist_emb80, float* ctx, int B, int T,
                           at_stream_t stream);

/* The same attention on the path the product runs (ref audiotoken/modeling_wav2vec2_bert.py:46-73; HF HubertAttention with dist_emb80 = NULL): k / v are
 * first written as the row-major fp16 pieces the fused q / k / v projection's epilogue produces and the distance embeddings as the fp16 pieces finalize()
 * prepares per layer (dist_max_abs = max |dist_emb80|: their power-of-two scale), both into kv_workspace (4 * ceil256(B * T) * heads * 64 * 2 + 24576
 * bytes), then w8 = 1 runs csrc/attention_f16x2_w8.hip (8-wave workgroups, 64-key tiles, LDS-DMA staging), w8 = 0 its round-3 twin
 * (csrc/attention_bf16x3.hip <SchemeF16x2, KVP>), w8 = -1 the default. status_dev nullable (bit 1 = fp16 range overflow; {flag, census} pair). */
int at_op_relpos_attention_kvp(const float* qkv, const float* attn_mask, const float* dist_emb80, float dist_max_abs, float* ctx, int B, int T, int heads, int w8,
                               void* kv_workspace, size_t kv_workspace_bytes, int32_t* status_dev, at_stream_t stream);

/* Conformer conv middle (HF modeling_wav2vec2_bert.py:212-222): causal depthwise k31 -> LayerNorm -> swish;
 * g [B*T][1024], w [31][1024]. */
int at_op_dwconv_ln_swish(const float* g, const float* w31x1024, const float* gamma, const float* beta, float* out, int B, int T,
                          at_stream_t stream);
/* the same op on the streaming kernel (csrc/dwconv_stream.hip); bit-identical to at_op_dwconv_ln_swish */
int at_op_dwconv_stream(const float* g, const float* w31x1024, const float* gamma, const float* beta, float* out, int B, int T,
                        at_stream_t stream);

/* VQ assign from precomputed dots [rows][C]: argmax_n -sqrt(max(|x|^2 + e2[n] - 2 dots, 0)), first index. */
int at_op_vq_argmax(const float* x, const float* dots, const float* e2, int16_t* out, int64_t rows, int D, int C, at_stream_t stream);
/* The same with the code rows given (codebook [C][D] fp32, round 5): codes whose approximate squared distance lies within (|x|^2 + e2[best]) 2^-17 of the
 * best are re-evaluated exactly — sum_k (x_k - e_k)^2 in float64 — and the smallest exact distance wins, ties to the lower index. What the two semantic
 * tokenizers run (option "vq_refine"); replaces the nearest-code step of vector_quantize_pytorch / torch.cdist + argmin (reference audiotoken/encoder.py:
 * 100-101,180-181), whose expanded fp32 form cancels when the centres sit in the data. */
int at_op_vq_argmax_refined(const float* x, const float* dots, const float* e2, const float* codebook, int16_t* out, int64_t rows, int D, int C, at_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* AUDIOTOKEN_HIP_H */
