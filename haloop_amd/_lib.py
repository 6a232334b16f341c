"""ctypes binding of libhalo.so (include/halo.h).  There is NO fallback: if the HIP library is
missing or a call fails, the product raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libhalo.so')

HALO_ABI_VERSION = 19
HALO_GEMM_RELU = 1
HALO_GEMM_GELU = 2
HALO_GEMM_ACCUM = 4
HALO_GEMM_GELU_ERF = 8
HALO_CTC_FULL_LATTICE = 1
HALO_CTC_FINITE_MIN = 2
HALO_CTC_NO_LEAD_BLANK_LOOP = 4
HALO_CTC_WRAP_SKIP = 8
HALO_STREAM_SUBSAMPLE = 1
HALO_STREAM_CLASSIFIER = 2
HALO_STREAM_LSTM_LAYER0 = 16
HALO_SUMSQ_PARTS = 1024
HALO_MATH_F32 = 0
HALO_MATH_BF16X3 = 1
HALO_MATH_BF16 = 2

_vp, _i, _l, _f, _u64, _u32, _sz = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_uint64, C.c_uint32, C.c_size_t

# name -> (restype, argtypes); one entry per function declared in include/halo.h
SIGNATURES = {
    'halo_abi_version': (_i, []),
    'halo_debug_read': (_i, [_vp, _sz, _i, _sz, _vp, _vp]),
    'halo_strerror': (C.c_char_p, [_i]),
    'halo_device_info': (_i, [_i, C.c_char_p, _i, C.POINTER(_i)]),
    'halo_set_math_mode': (_i, [_i]),
    'halo_get_math_mode': (_i, []),
    'halo_set_scratch': (_i, [_vp, _sz]),
    'halo_set_lstm_fusion': (_i, [_i]),
    'halo_dropout_fwd': (_i, [_vp, _vp, _sz, _f, _u64, _u32, _u32, _vp, _vp]),
    'halo_counter_inc': (_i, [_vp, _vp]),
    'halo_gemm_f32': (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _f, _u64, _u32, _u32, _vp, _vp]),
    'halo_split_image_bytes': (_sz, [_i, _i]),
    'halo_split_image': (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    'halo_layernorm_image': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    'halo_layernorm_bf16': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    'halo_cast_bf16': (_i, [_vp, _vp, _sz, _vp]),
    'halo_gelu_bf16': (_i, [_vp, _vp, _sz, _i, _vp]),
    'halo_gelu_bwd_bf16': (_i, [_vp, _vp, _vp, _sz, _i, _vp]),
    'halo_gemm_tn_bf16_group': (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp]),
    'halo_gemm_tn_rows_supported': (_i, [_i, _vp, _vp, _i]),
    'halo_gemm_tn_rows_preferred': (_i, [_i, _vp, _vp, _i]),
    'halo_gemm_tn_rows_group': (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    'halo_gelu_b16': (_i, [_vp, _vp, _sz, _i, _vp]),
    'halo_gelu_bwd_b16': (_i, [_vp, _vp, _vp, _sz, _i, _vp]),
    'halo_dx_alloc': (_i, [_sz, _vp, _vp]),
    'halo_dx_open': (_i, [_vp, _vp]),
    'halo_dx_close': (_i, [_vp]),
    'halo_dx_free': (_i, [_vp]),
    'halo_dx_push': (_i, [_vp, _sz, _sz, _i, _vp, _sz, _i, _i, _vp]),
    'halo_dx_signal': (_i, [_vp, _sz, _i, _i, _u32, _vp]),
    'halo_dx_wait': (_i, [_vp, _i, _i, _u32, _vp]),
    'halo_dx_reduce': (_i, [_vp, _vp, _sz, _i, _i, _f, _vp]),
    'halo_attention_fwd_b16': (_i, [_vp, _l, _l, _vp, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'halo_attention_bwd_b16': (_i, [_vp, _l, _l, _vp, _vp, _l, _l, _vp, _vp, _l, _l, _vp, _vp, _vp, _vp, _vp, _l, _l, _i, _i, _i, _i, _i, _i, _vp]),
    'halo_gemm_rows_supported': (_i, [_i, _i, _i]),
    'halo_gemm_rows': (_i, [_vp, _vp, _l, _vp, _i, _i, _i, _vp, _l, _vp, _l, _vp, _l, _vp]),
    'halo_gemm_rows_gelu': (_i, [_vp, _vp, _l, _vp, _i, _i, _i, _vp, _vp, _l, _i, _vp]),
    'halo_gemm_rows_ce_workspace_bytes': (_sz, [_i, _i]),
    'halo_gemm_rows_ce': (_i, [_vp, _vp, _l, _vp, _i, _i, _i, _vp, _l, _vp, _vp, _vp, _vp, _l, _vp]),
    'halo_cross_entropy_bwd_bf16': (_i, [_vp, _vp, _vp, _vp, _l, _i, _i, _l, _l, _vp]),
    'halo_gemm_split_ce_workspace_bytes': (_sz, [_i, _i]),
    'halo_gemm_split_ce': (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _l, _vp, _vp, _vp, _vp]),
    'halo_image_pair': (_i, [_vp, _vp, _i, _i, _l, _l, _i, _vp, _vp, _vp]),
    'halo_cross_entropy_bwd_images': (_i, [_vp, _vp, _vp, _vp, _l, _i, _i, _l, _l, _vp, _vp, _vp]),
    'halo_gemm_split': (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _i, _f, _u64, _u32, _u32, _vp, _vp]),
    'halo_gemm_split_residual': (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _f, _u64, _u32, _u32, _vp, _vp]),
    'halo_image_pairs': (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'halo_gemm_split_io': (_i, [_vp, _vp, _vp, _l, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _l, _vp, _i, _vp, _vp, _i, _vp]),
    'halo_gemm_tn_bf16': (_i, [_vp, _l, _vp, _l, _i, _i, _i, _vp, _i, _i, _vp]),
    'halo_subsample_col_bytes': (_sz, [_i] * 6),
    'halo_subsample_fwd': (_i, [_vp] * 5 + [_i] * 7 + [_f, _u64, _u32, _vp, _vp]),
    'halo_subsample_bwd': (_i, [_vp] * 6 + [_i] * 7 + [_f, _vp]),
    'halo_subsample_bwd_slabs': (_i, [_vp, _i] + [_vp] * 5 + [_i] * 7 + [_f, _vp]),
    'halo_lstm_reserve_bytes': (_sz, [_i] * 5),
    'halo_lstm_bwd_workspace_bytes': (_sz, [_i] * 5),
    'halo_lstm_fwd': (_i, [_vp] * 8 + [_l, _l, _i, _vp, _vp, _vp] + [_i] * 5 + [_f, _u64, _u32, _vp, _vp]),
    'halo_lstm_bwd': (_i, [_vp] * 4 + [_l, _l, _i] + [_vp] * 9 + [_i] * 7 + [_f, _u64, _u32, _vp, _vp]),
    'halo_set_lstm_persistent': (_i, [_i]),
    'halo_set_lstm_persistent_images': (_i, [_i]),
    'halo_lstm_persistent_eligible': (_i, [_i, _i]),
    'halo_set_lstm_persistent2': (_i, [_i]),
    'halo_set_lstm_interleave': (_i, [_i]),
    'halo_set_gemm256': (_i, [_i]),
    'halo_set_lstm_bwd_mid_event': (_i, [_vp]),
    'halo_lstm_bwd_mid_event_recorded': (_i, []),
    'halo_set_lstm_expect_backward': (_i, [_i]),
    'halo_set_lstm_dx_slabs': (_i, [_i]),
    'halo_set_defer_small_jobs': (_i, [_i]),
    'halo_set_grad_sumsq': (_i, [_vp, _i]),
    'halo_grad_sumsq_state': (_i, [_vp, _vp]),
    'halo_flush_small_jobs': (_i, [_vp]),
    'halo_lstm_dx_slabs_left': (_i, []),
    'halo_set_lstm_weights_stamp': (_i, [_u64]),
    'halo_lstm_persistent2_eligible': (_i, [_i, _i, _i, _i]),
    'halo_lstm_status_offset': (_sz, [_i] * 6),
    'halo_lstm_persist_stamps': (_i, [_vp]),
    'halo_lstm_chain_events': (_i, [_vp, _vp]),
    'halo_lstm_chain_info': (_i, [_i, C.POINTER(_i), C.c_char_p, _i]),
    'halo_log_softmax_fwd': (_i, [_vp, _vp, _i, _i, _vp]),
    'halo_log_softmax_bwd': (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    'halo_colsum': (_i, [_vp, _i, _i, _i, _vp, _vp]),
    'halo_ctc_fwd': (_i, [_vp, _l, _l, _i, _i, _i, _vp, _l, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    'halo_ctc_bwd': (_i, [_vp, _l, _l, _i, _i, _i, _vp, _l, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _l, _vp]),
    'halo_ctc_prepare': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'halo_ctc_mean_loss': (_i, [_vp, _vp, _i, _vp, _vp]),
    'halo_ctc_head_supported': (_i, [_i, _i, _i, _i]),
    'halo_ctc_head_greedy': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'halo_ctc_head_workspace_bytes': (_sz, [_i, _i, _i]),
    'halo_ctc_head_fwd': (_i, [_vp, _vp, _vp, _f, _u64, _u32, _u32, _vp, _vp, _i, _i, _i, _vp, _l, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                               _vp, _i, _i, _i, _i, _vp]),
    'halo_ctc_head_bwd': (_i, [_vp, _vp, _f, _u64, _u32, _u32, _vp, _vp, _vp, _l, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                               _i, _i, _i, _i, _vp]),
    'halo_ctc_head_train_workspace_bytes': (_sz, [_i, _i, _i]),
    'halo_ctc_head_train_ticket_words': (_sz, [_i, _i]),
    'halo_ctc_head_train': (_i, [_vp, _vp, _vp, _f, _u64, _u32, _u32, _vp, _vp, _i, _i, _i, _vp, _l, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                 _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'halo_ctc_greedy': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'halo_set_beam_vector_chunk': (_i, [_i]),
    'halo_logaddexp_aten': (_i, [_vp, _vp, _vp, _sz, _vp]),
    'halo_ctc_beam_workspace_bytes': (_sz, [_i] * 4),
    'halo_ctc_beam': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'halo_topk_f32': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'halo_embed_fwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'halo_layernorm_fwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    'halo_attention_causal_fwd': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'halo_cross_entropy_fwd': (_i, [_vp, _vp, _vp, _i, _i, _l, _l, _vp]),
    'halo_attention_fwd': (_i, [_vp, _l, _l, _vp, _vp, _l, _l, _vp, _l, _l, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'halo_attention_fwd_strided': (_i, [_vp, _l, _l, _l, _vp, _vp, _l, _l, _l, _vp, _l, _l, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp,
                                        _f, _u64, _u32, _u32, _vp, _vp]),
    'halo_kv_cache_store_f32': (_i, [_vp, _l, _l, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'halo_rope_table': (_i, [_vp, _vp, _i, _i, _f, _vp]),
    'halo_rope_interleaved': (_i, [_vp, _l, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp]),
    'halo_kv_cache_store': (_i, [_vp, _l, _l, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'halo_attention_decode': (_i, [_vp, _l, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'halo_attention_decode_step': (_i, [_vp, _vp, _vp, _l, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    'halo_logprob_max': (_i, [_vp, _l, _i, _i, _vp, _vp, _vp, _vp]),
    'halo_greedy_update': (_i, [_vp, _vp, _vp, _vp, _l, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    'halo_fbank_frames': (_i, [_vp, _l, _i, _i, _i, _f, _i, _vp, _vp, _i, _vp]),
    'halo_fbank_power': (_i, [_vp, _i, _i, _vp, _i, _vp]),
    'halo_fbank_log': (_i, [_vp, _l, _f, _vp]),
    'halo_fbank_spectrum_mel': (_i, [_vp, _i, _i, _vp, _vp, _i, _f, _vp, _vp]),
    'halo_star_ctc_workspace_bytes': (_sz, [_i, _i, _i]),
    'halo_star_ctc_fwd': (_i, [_vp, _l, _l, _i, _i, _i, _vp, _i, _vp, _vp, _f, _vp, _vp, _vp]),
    'halo_star_ctc_bwd': (_i, [_vp, _l, _l, _i, _i, _i, _vp, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp]),
    'halo_transducer_workspace_bytes': (_sz, [_i, _i, _i]),
    'halo_transducer_fwd': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'halo_transducer_bwd': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'halo_decode_image_bytes': (_sz, [_i, _i]),
    'halo_decode_image': (_i, [_vp, _i, _i, _l, _vp, _vp]),
    'halo_decode_linear_supported': (_i, [_i, _i]),
    'halo_decode_linear': (_i, [_vp, _l, _i, _i, _vp, _f, _vp, _i, _vp, _l, _i, _vp]),
    'halo_decode_linear_pair': (_i, [_vp, _vp, _l, _i, _i, _vp, _f, _vp, _i, _vp, _vp, _vp, _l, _i, _vp]),
    'halo_decode_attention_pair': (_i, [_vp, _l, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _l, _vp]),
    'halo_decode_memory_caches': (_i, [_vp, _l, _i, _vp, _i, _i, _i, _i, _vp]),
    'halo_decode_token': (_i, [_vp, _l, _i, _i, _vp, _l, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    'halo_attention_bwd': (_i, [_vp, _l, _l, _vp, _vp, _l, _l, _vp, _vp, _l, _l, _vp, _vp, _vp, _l, _l, _vp, _vp, _l, _l,
                                _i, _i, _i, _i, _i, _i, _vp, _f, _u64, _u32, _u32, _vp, _vp]),
    'halo_attention_masked': (_i, [_vp, _vp, _vp, _vp, _l, _l, _l, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'halo_attention_fwd_bf16': (_i, [_vp, _l, _l, _vp, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp, _i, _i, _i, _i, _i, _i, _vp, _f, _u64, _u32, _u32,
                                     _vp, _vp]),
    'halo_attention_bwd_bf16': (_i, [_vp, _l, _l, _vp, _vp, _l, _l, _vp, _vp, _l, _l, _vp, _vp, _vp, _vp, _vp, _l, _l, _i, _i, _i, _i, _i, _i, _vp,
                                     _f, _u64, _u32, _u32, _vp, _vp]),
    'halo_layernorm_bwd_workspace_bytes': (_sz, [_i, _i]),
    'halo_layernorm_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    'halo_layernorm_bwd_bf16': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    'halo_layernorm_bwd_b16': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    'halo_gelu_fwd': (_i, [_vp, _vp, _sz, _i, _vp]),
    'halo_gelu_bwd': (_i, [_vp, _vp, _vp, _sz, _i, _vp]),
    'halo_cross_entropy_fwd_lse': (_i, [_vp, _vp, _vp, _vp, _i, _i, _l, _l, _vp]),
    'halo_cross_entropy_bwd': (_i, [_vp, _vp, _vp, _vp, _l, _i, _i, _l, _l, _vp]),
    'halo_embed_bwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'halo_add_rows_bcast': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'halo_im2col_cl': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'halo_col2im_cl': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'halo_dwconv1d_cl': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'halo_dwconv1d_cl_bwd_workspace_bytes': (_sz, [_i, _i]),
    'halo_dwconv1d_cl_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'halo_tape_batch': (_i, [_vp, _i, _l, _i, _i, _l, _i, _l, _vp, _vp]),
    'halo_lm_batch_u16': (_i, [_vp, _l, _vp, _i, _i, _i, _vp, _vp, _vp]),
    'halo_scale_add': (_i, [_vp, _vp, _f, _f, _sz, _vp]),
    'halo_cast_f32_bf16': (_i, [_vp, _vp, _sz, _vp]),
    'halo_cast_bf16_f32': (_i, [_vp, _vp, _f, _sz, _vp]),
    'halo_scale_add_guarded': (_i, [_vp, _vp, _f, _f, _sz, _vp, _vp]),
    'halo_clip_coef_step': (_i, [_vp, _i, _f, _vp, _vp, _vp, _vp]),
    'halo_adamw_ranges_dev': (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _f, _vp, _f, _f, _f, _vp, _vp, _vp]),
    'halo_ctx_create': (_vp, []),
    'halo_ctx_destroy': (None, [_vp]),
    'halo_ctx_use': (_i, [_vp]),
    'halo_set_status_word': (_i, [_vp]),
    'halo_debug_mute_workgroup': (_i, [_i]),
    'halo_debug_mfma_clock': (_i, [_vp, _vp, _i, _i, _i, _u32, _vp]),
    'halo_sumsq': (_i, [_vp, _sz, _vp, _vp]),
    'halo_sumsq_ranges': (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    'halo_pack_ranges_bf16': (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    'halo_expand_ranges_bf16': (_i, [_vp, _i, _vp, _vp, _i, _i, _vp, _vp]),
    'halo_clip_coef': (_i, [_vp, _i, _f, _vp, _vp, _vp]),
    'halo_adamw_multi_tensor_bytes': (_sz, []),
    'halo_adamw_multi_chunk': (_u32, []),
    'halo_adamw_multi_max_tensors': (_i, []),
    'halo_adamw_multi': (_i, [_vp, _vp, _i, _vp, _i, _f, _f, _f, _f, _i, _vp, _vp]),
    'halo_adamw_ranges': (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _i, _vp, _vp]),
    'halo_adamw': (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _i, _vp, _vp]),
}

_lib = None


class HaloError(RuntimeError):
    pass


def lib():
    """The loaded library; raises (never falls back) when it is missing or mismatched."""
    global _lib
    if _lib is None:
        # torch ships its own libamdhip64: it must be in the process BEFORE libhalo.so resolves the same soname, or the
        # library binds a second HIP runtime and every launch on torch's streams fails (HALO_ELAUNCH)
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise HaloError(f'{LIB_PATH} not found: build it with `make -C haloop_amd/csrc` '
                            '(or __graft_entry__.build()); haloop_amd has no CPU fallback')
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)            # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if handle.halo_abi_version() != HALO_ABI_VERSION:
            raise HaloError('libhalo.so ABI version mismatch: rebuild it')
        _lib = handle
        mode = os.environ.get('HALO_MATH')
        if mode:
            set_math_mode(mode)
    return _lib


def set_math_mode(mode):
    """'f32' (exact-f32 MFMA), 'bf16x3' (split-bf16, three MFMAs per product: fp32-grade) or 'bf16' (operands rounded to
    bf16, one MFMA per product: the reference's bf16-autocast arithmetic) for the dense products (include/halo.h)."""
    global _mode
    code = {'f32': HALO_MATH_F32, 'bf16x3': HALO_MATH_BF16X3, 'bf16': HALO_MATH_BF16}[mode]
    check(lib().halo_set_math_mode(code), 'halo_set_math_mode')
    _mode = mode


def set_lstm_fusion(on):
    """Layer-diagonal fused schedule for multi-layer LSTMs (off by default; see include/halo.h)."""
    check(lib().halo_set_lstm_fusion(int(bool(on))), 'halo_set_lstm_fusion')


def set_lstm_persistent(on):
    """Weight-resident persistent LSTM recurrence (one launch per layer and direction) on / off (include/halo.h)."""
    check(lib().halo_set_lstm_persistent(int(bool(on))), 'halo_set_lstm_persistent')


def set_lstm_bwd_mid_event(event):
    """event: a torch.cuda.Event that has been recorded at least once (so that its handle exists), or None.  halo_lstm_bwd records it
    behind the launch that stores the top LSTM layer's weight gradients (include/halo.h)."""
    check(lib().halo_set_lstm_bwd_mid_event(None if event is None else C.c_void_p(event.cuda_event)), 'halo_set_lstm_bwd_mid_event')


def set_gemm256(on):
    check(lib().halo_set_gemm256(1 if on else 0), 'halo_set_gemm256')


def set_lstm_interleave(on):
    """Two batch tiles per workgroup, interleaved, in the two-layer launches of a batch larger than one launch holds (include/halo.h)."""
    check(lib().halo_set_lstm_interleave(int(bool(on))), 'halo_set_lstm_interleave')


def set_lstm_persistent2(on):
    """Both layers of a 2-layer LSTM in one persistent launch per direction (bf16 mode; include/halo.h) on / off."""
    check(lib().halo_set_lstm_persistent2(int(bool(on))), 'halo_set_lstm_persistent2')


# The optimizer kernels write parameters through raw pointers: torch's per-tensor version counters do not see it.  Every Python-side
# path that lets such a kernel run (ops.adamw*, a replay of a captured training step) bumps this counter, and caches of data derived
# from weights that outlive a call (infer.LstmCtcRecognizer's packed LSTM weights) carry it in their stamp.
_WEIGHTS_EPOCH = [0]


def bump_weights_epoch():
    _WEIGHTS_EPOCH[0] += 1


def weights_epoch():
    return _WEIGHTS_EPOCH[0]


def set_lstm_expect_backward(on):
    """Whether a backward follows the LSTM forwards issued next (include/halo.h): inference switches it off."""
    check(lib().halo_set_lstm_expect_backward(int(bool(on))), 'halo_set_lstm_expect_backward')


def set_lstm_persistent_images(on):
    """The persistent LSTM backward emitting the gate gradients' GEMM operand images itself on / off (include/halo.h)."""
    check(lib().halo_set_lstm_persistent_images(int(bool(on))), 'halo_set_lstm_persistent_images')


def lstm_chain_events(ev_begin, ev_end):
    """Measurement hook (include/halo.h): torch.cuda.Events (already recorded once, so their handles exist) that the library
    records around every LSTM recurrent chain; (None, None) clears."""
    h0 = None if ev_begin is None else ev_begin.cuda_event
    h1 = None if ev_end is None else ev_end.cuda_event
    check(lib().halo_lstm_chain_events(h0, h1), 'halo_lstm_chain_events')


def lstm_chain_info(direction):
    n = _i(0)
    buf = C.create_string_buffer(128)
    check(lib().halo_lstm_chain_info(1 if direction == 'bwd' else 0, C.byref(n), buf, 128), 'halo_lstm_chain_info')
    return {'launches': n.value, 'kernel': buf.value.decode()}


_mode = None       # the library's arithmetic mode, mirrored here: it only changes through set_math_mode, and the Linear helpers
                   # ask for it on every call


def get_math_mode():
    global _mode
    if _mode is None:
        _mode = {HALO_MATH_F32: 'f32', HALO_MATH_BF16X3: 'bf16x3', HALO_MATH_BF16: 'bf16'}[lib().halo_get_math_mode()]
    return _mode


_scratch = None


def lend_scratch(nbytes=64 << 20, device=None):
    """Allocate (once, with torch) and lend the library its split-K scratch buffer."""
    global _scratch
    import torch
    if _scratch is None or _scratch.numel() < nbytes:
        _scratch = torch.empty(nbytes, dtype=torch.uint8, device=device or 'cuda')
        check(lib().halo_set_scratch(_scratch.data_ptr(), _scratch.numel()), 'halo_set_scratch')
    return _scratch


class Context:
    """A caller-owned settings record (include/halo.h, "Contexts"): created as a copy of the calling thread's current settings;
    ``with ctx:`` makes it the record this thread's halo_* calls read and halo_set_* calls write, and restores the previous
    selection on exit.  Two trainers (or a trainer and a recognizer) on different threads then do not share switches."""
    _current = __import__('threading').local()

    def __init__(self):
        self.handle = lib().halo_ctx_create()
        if not self.handle:
            raise HaloError('halo_ctx_create failed')
        self._prev = []

    def use(self):
        global _mode
        check(lib().halo_ctx_use(self.handle), 'halo_ctx_use')
        Context._current.ctx = self
        _mode = None                      # re-read the arithmetic mode of the selected record

    def __enter__(self):
        self._prev.append(getattr(Context._current, 'ctx', None))
        self.use()
        return self

    def __exit__(self, *exc):
        global _mode
        prev = self._prev.pop()
        check(lib().halo_ctx_use(prev.handle if prev is not None else None), 'halo_ctx_use')
        Context._current.ctx = prev
        _mode = None
        return False

    def close(self):
        if self.handle:
            lib().halo_ctx_destroy(self.handle)
            self.handle = None


_status_tensor = None


def set_status_word(t):
    """Lend the library a device int32/uint32 tensor of one element as its sticky status word (None: none); include/halo.h.
    The tensor is kept alive here for as long as it is the registered word."""
    global _status_tensor
    check(lib().halo_set_status_word(ptr(t)), 'halo_set_status_word')
    _status_tensor = t


def check(rc, what):
    if rc != 0:
        raise HaloError(f'{what} failed: {lib().halo_strerror(rc).decode()} ({rc})')


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def ptr_array(tensors):
    """Host array of device pointers (kept alive by the caller for the duration of the call)."""
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr
