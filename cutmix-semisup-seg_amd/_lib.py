"""
ctypes binding of libcutmixseg_hip.so (C ABI: include/cutmixseg.h).

There is deliberately NO fallback: if the shared object has not been built (python cutmix-semisup-seg_amd/build.py,
or __graft_entry__.build()) importing this module raises, and every op in the package fails with it.
"""
import ctypes as C
import os

# PyTorch ships its own libamdhip64.so.7; it must be the HIP runtime instance this library binds to (same SONAME), or
# kernels would be launched through a second, device-less runtime. Importing torch first guarantees that.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libcutmixseg_hip.so')

if not os.path.exists(LIB_PATH):
    raise ImportError('libcutmixseg_hip.so not found at {} -- build it with '
                      '`python cutmix-semisup-seg_amd/build.py` (hipcc, gfx950); there is no CPU fallback'.format(LIB_PATH))

lib = C.CDLL(LIB_PATH)

F32, BF16 = 0, 1
LABEL_U8, LABEL_I64 = 0, 1
LOSS_IDS = {'var': 0, 'logits_var': 1, 'logits_smoothl1': 2, 'bce': 3, 'kld': 4}
MODE_MIX, MODE_CUT = 0, 1
OPT_CHUNK = 2048
AUG_PARAMS = 24

c_void_p, c_int, c_float, c_size_t = C.c_void_p, C.c_int, C.c_float, C.c_size_t


class ConsistencyDesc(C.Structure):
    _fields_ = [('l_stu', c_void_p), ('l_tea0', c_void_p), ('l_tea1', c_void_p), ('ranges', c_void_p),
                ('mask', c_void_p), ('um0', c_void_p), ('um1', c_void_p),
                ('n', c_int), ('c', c_int), ('h', c_int), ('w', c_int), ('H', c_int), ('W', c_int),
                ('align_corners', c_int), ('n_boxes', c_int), ('invert', c_int), ('mode', c_int),
                ('loss_fn', c_int), ('conf_thresh', c_float), ('conf_per_pixel', c_int)]


class CeDesc(C.Structure):
    _fields_ = [('logits', c_void_p), ('labels', c_void_p), ('label_dtype', c_int), ('ignore_index', c_int),
                ('n', c_int), ('c', c_int), ('h', c_int), ('w', c_int), ('H', c_int), ('W', c_int),
                ('align_corners', c_int)]


class ParamSegment(C.Structure):
    _fields_ = [('offset', C.c_uint64), ('count', C.c_uint64), ('k_updates', C.c_int32), ('lr_group', C.c_int32)]


class OptimDesc(C.Structure):
    _fields_ = [('param', c_void_p), ('grad', c_void_p), ('slot0', c_void_p), ('slot1', c_void_p),
                ('ema_param', c_void_p), ('param_bf16', c_void_p), ('ema_bf16', c_void_p),
                ('segments', c_void_p), ('chunk_seg', c_void_p), ('chunk_off', c_void_p), ('n_chunks', C.c_uint32),
                ('lrs', c_void_p), ('step_count', c_void_p),
                ('grad_scale', c_float), ('ema_alpha', c_float), ('ema_one_minus_alpha', c_float),
                ('beta1', C.c_double), ('beta2', C.c_double), ('eps', C.c_double),
                ('momentum', c_float), ('weight_decay', c_float), ('nesterov', c_int)]


class ConvDesc(C.Structure):
    _fields_ = [('x', c_void_p), ('w', c_void_p), ('y', c_void_p), ('y32', c_void_p), ('scale', c_void_p),
                ('bias', c_void_p), ('res', c_void_p), ('mask_src', c_void_p),
                ('n', c_int), ('h', c_int), ('w_in', c_int), ('cin', c_int),
                ('ho', c_int), ('wo', c_int), ('cout', c_int), ('cout_real', c_int), ('ntaps', c_int),
                ('tap_dy', c_int * 18), ('tap_dx', c_int * 18), ('stride', c_int),
                ('out_h', c_int), ('out_w', c_int), ('out_stride', c_int), ('relu', c_int), ('mode', c_int),
                ('tile', c_int), ('ksplit', c_int), ('zeros', c_void_p), ('variant', c_int), ('zeros_bytes', c_int),
                ('workspace', c_void_p), ('workspace_bytes', C.c_longlong), ('mask_bits_out', c_void_p), ('mask_bits', c_void_p),
                ('stats_out', c_void_p), ('stats_rows_per_group', c_int), ('mask_gates_res', c_int),
                ('bstats_u', c_void_p), ('bstats_bits', c_void_p), ('bstats_mean', c_void_p), ('bstats_rstd', c_void_p)]


class WgradDesc(C.Structure):
    _fields_ = [('du', c_void_p), ('x', c_void_p), ('dw', c_void_p), ('scale', c_void_p),
                ('n', c_int), ('h', c_int), ('w_in', c_int), ('cin', c_int), ('ho', c_int), ('wo', c_int),
                ('cout', c_int), ('cout_real', c_int), ('ntaps', c_int), ('tap_dy', c_int * 18), ('tap_dx', c_int * 18),
                ('stride', c_int), ('ksplit', c_int), ('w', c_void_p), ('wdot', c_void_p), ('dbeta', c_void_p),
                ('dw_cout', c_int), ('workspace', c_void_p), ('workspace_bytes', C.c_longlong), ('wg_target', c_int)]


class BnOp(C.Structure):
    _fields_ = [('what', c_int), ('dtype', c_int), ('c', c_int), ('relu', c_int),
                ('x', c_void_p), ('res', c_void_p), ('y', c_void_p), ('dy', c_void_p), ('dx', c_void_p), ('dres', c_void_p),
                ('sums', c_void_p), ('gamma', c_void_p), ('beta', c_void_p), ('mean', c_void_p), ('rstd', c_void_p),
                ('scale', c_void_p), ('shift', c_void_p), ('running_mean', c_void_p), ('running_var', c_void_p),
                ('counter', c_void_p), ('clear_a', c_void_p), ('clear_b', c_void_p), ('ws', c_void_p), ('count', C.c_double), ('n_pixels', C.c_ulonglong), ('eps', c_float),
                ('momentum', c_float), ('groups', c_int), ('reserved', c_int), ('mask_bits', c_void_p)]


class AugmentDesc(C.Structure):
    _fields_ = [('src', c_void_p), ('src_labels', c_void_p), ('out0', c_void_p), ('out1', c_void_p),
                ('out_labels', c_void_p), ('out_mask', c_void_p), ('params', c_void_p),
                ('mean', c_float * 3), ('std_', c_float * 3),
                ('n', c_int), ('hs', c_int), ('ws', c_int), ('h', c_int), ('w', c_int), ('out_dtype', c_int)]


class PackItem(C.Structure):
    _fields_ = [('src', c_void_p), ('dst', c_void_p), ('scale', c_void_p),
                ('ntaps', c_int), ('cout', c_int), ('cin', c_int), ('first_block', c_int)]


class WfinishItem(C.Structure):
    _fields_ = [('scratch', c_void_p), ('grad', c_void_p), ('w', c_void_p), ('scale', c_void_p), ('wdot', c_void_p),
                ('ntaps', c_int), ('cout', c_int), ('cin', c_int), ('first_block', c_int)]


_P = C.POINTER


def _proto(name, restype, argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = argtypes
    return fn


# every symbol include/cutmixseg.h declares (tests/test_abi.py checks the header against this table)
PROTOTYPES = {
    'cms_version': (c_int, []),
    'cms_last_error': (C.c_char_p, []),
    'cms_device_info': (c_int, [_P(c_int), C.c_char_p, c_size_t]),
    'cms_boxmask_rasterize': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'cms_cutmix_paste': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                 c_int, c_void_p]),
    'cms_cutmix_paste_mask': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                      c_void_p]),
    'cms_consistency_workspace_bytes': (c_size_t, [_P(ConsistencyDesc)]),
    'cms_consistency_fwd': (c_int, [_P(ConsistencyDesc), c_void_p, c_void_p, c_void_p]),
    'cms_consistency_finalize': (c_int, [c_void_p, c_void_p, c_float, c_int, c_float, c_float, c_void_p, c_void_p]),
    'cms_consistency_bwd': (c_int, [_P(ConsistencyDesc), c_void_p, c_void_p, c_void_p]),
    'cms_consistency_fused_supported': (c_int, [_P(ConsistencyDesc)]),
    'cms_consistency_fwd_bwd': (c_int, [_P(ConsistencyDesc), c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    'cms_scale_by_scalar': (c_int, [c_void_p, C.c_longlong, c_void_p, c_int, c_float, c_void_p]),
    'cms_ce_workspace_bytes': (c_size_t, [_P(CeDesc)]),
    'cms_ce_fwd': (c_int, [_P(CeDesc), c_void_p, c_void_p, c_void_p]),
    'cms_ce_finalize': (c_int, [c_void_p, c_float, c_void_p, c_void_p]),
    'cms_ce_bwd': (c_int, [_P(CeDesc), c_void_p, c_void_p, c_void_p]),
    'cms_ce_fused_supported': (c_int, [_P(CeDesc)]),
    'cms_ce_fwd_bwd': (c_int, [_P(CeDesc), c_void_p, c_void_p, c_void_p, c_void_p]),
    'cms_upsample_bilinear_fwd': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                          c_void_p]),
    'cms_upsample_bilinear_bwd': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                          c_void_p]),
    'cms_ema_flat': (c_int, [c_void_p, c_void_p, c_size_t, c_float, c_float, c_void_p, c_void_p]),
    'cms_adam_ema_step': (c_int, [_P(OptimDesc), c_void_p]),
    'cms_sgd_ema_step': (c_int, [_P(OptimDesc), c_void_p]),
    'cms_increment_counter': (c_int, [c_void_p, c_void_p]),
    'cms_bn_fold': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    'cms_argmax_confusion': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                     c_int, c_void_p, c_void_p, c_void_p]),
    'cms_confusion': (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p, c_void_p]),
    'cms_conv_igemm': (c_int, [_P(ConvDesc), c_void_p]),
    'cms_conv_igemm_workspace_bytes': (C.c_longlong, []),
    'cms_conv_igemm_route': (c_int, [_P(ConvDesc)]),
    'cms_conv_igemm_stats_tile_rows': (c_int, [_P(ConvDesc)]),
    'cms_bn_finalize_tiles': (c_int, [c_void_p, c_int, c_size_t, c_int, c_int, c_void_p, c_void_p, c_float, c_float, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'cms_frozen_bn_act_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_int, c_void_p]),
    'cms_bn_apply_groups_bits': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_size_t, c_int, c_int,
                                         c_void_p, c_void_p]),
    'cms_bn_reduce_ws_bits': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int,
                                      c_void_p, c_void_p]),
    'cms_bn_bwd_apply_groups_bits': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                             c_void_p, c_void_p, C.c_double, c_size_t, c_int, c_int, c_void_p]),
    'cms_bn_bwd_sums_tiles': (c_int, [c_void_p, c_int, c_size_t, c_int, c_int, c_void_p, c_void_p]),
    'cms_conv_set_trace': (c_int, [c_void_p, c_int]),
    'cms_conv_set_wgrad8': (c_int, [c_int]),
    'cms_loss_set_deterministic': (c_int, [c_int]),
    'cms_conv_wgrad_uses_wgrad8': (c_int, [_P(WgradDesc)]),
    'cms_conv_wgrad': (c_int, [_P(WgradDesc), c_void_p]),
    'cms_conv_wgrad_workspace_bytes': (C.c_longlong, [_P(WgradDesc)]),
    'cms_conv_wgrad_group_kind': (c_int, [_P(WgradDesc)]),
    'cms_conv_wgrad_group_bytes': (C.c_longlong, [c_int]),
    'cms_conv_wgrad_group_pack': (c_int, [_P(WgradDesc), c_int, c_int, c_void_p, C.c_longlong, _P(c_int)]),
    'cms_conv_wgrad_group_run': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    'cms_conv_pack_transpose': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'cms_conv_pack_transpose_batch': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    'cms_conv_pack_transpose_batch64': (c_int, [c_void_p, c_int, c_int, c_void_p]),
    'cms_augment_batch': (c_int, [_P(AugmentDesc), c_void_p]),
    'cms_augment_luma': (c_int, [_P(AugmentDesc), c_void_p, c_void_p]),
    'cms_bn_reduce': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int,
                              c_void_p]),
    'cms_bn_finalize': (c_int, [c_void_p, C.c_double, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'cms_bn_finalize_ex': (c_int, [c_void_p, C.c_double, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p,
                           c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'cms_bn_workspace_bytes': (c_size_t, [c_size_t, c_int, c_int]),
    'cms_bn_reduce_ws': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int,
                                 c_int, c_void_p, c_void_p]),
    'cms_bn_stats': (c_int, [c_void_p, c_int, c_size_t, c_int, c_int, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p,
                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'cms_bn_apply_groups': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_size_t, c_int, c_int,
                                    c_void_p]),
    'cms_bn_bwd_apply_groups': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p, C.c_double, c_size_t, c_int, c_int, c_void_p]),
    'cms_bn_apply': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_size_t, c_int, c_void_p]),
    'cms_bn_bwd_apply': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, C.c_double, c_size_t, c_int, c_void_p]),
    'cms_channel_copy': (c_int, [c_void_p, c_size_t, c_void_p, c_size_t, c_size_t, c_int, c_int, c_size_t, c_void_p]),
    'cms_add_n': (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_int, c_void_p]),
    'cms_channel_sum': (c_int, [c_void_p, c_int, c_size_t, c_int, c_void_p, c_void_p]),
    'cms_wgrad_finish_pack': (c_int, [_P(WfinishItem), c_int]),
    'cms_wgrad_finish_run': (c_int, [c_void_p, c_int, c_int, c_void_p]),
    'cms_program_add_channel_sum': (c_int, [c_void_p, c_void_p, c_int, c_size_t, c_int, c_void_p, c_int, c_int]),
    'cms_program_add_wgrad_finish': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int]),
    'cms_rows_reduce': (c_int, [c_void_p, c_size_t, c_int, c_size_t, c_int, c_int, c_void_p, c_float, c_void_p]),
    'cms_upsample_nhwc': (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                  c_void_p]),
    'cms_aspp_gather_fwd': (c_int, [c_void_p, c_void_p, c_void_p, _P(c_int), _P(c_int), c_int, c_int, c_int, c_int, c_int,
                                    c_int, c_void_p]),
    'cms_aspp_spread_bwd': (c_int, [c_void_p, c_void_p, c_int, _P(c_int), _P(c_int), c_int, c_int, c_int, c_int, c_int,
                                    c_int, c_void_p]),
    'cms_stem_pack_weights': (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    'cms_stem_out_hw': (c_int, [c_int, c_int, _P(c_int), _P(c_int), _P(c_int), _P(c_int)]),
    'cms_stem_fwd': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                             c_void_p]),
    'cms_maxpool3x3s2_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'cms_maxpool3x3s2_relu_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                          c_int, c_void_p]),
    'cms_stem_wgrad': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'cms_stem_wgrad_workspace_bytes': (C.c_longlong, [c_int, c_int, c_int, c_int, c_int]),
    'cms_stem_wgrad_ws': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                          C.c_longlong, c_void_p]),
    'cms_stem_dgrad': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'cms_program_create': (c_int, [_P(c_void_p)]),
    'cms_program_destroy': (c_int, [c_void_p]),
    'cms_program_add_conv': (c_int, [c_void_p, _P(ConvDesc), c_int, c_int, c_int]),
    'cms_program_add_wgrad': (c_int, [c_void_p, _P(WgradDesc), c_int, c_int, c_int]),
    'cms_program_add_wgrad_group': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int]),
    'cms_program_add_memset': (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int]),
    'cms_program_add_sync': (c_int, [c_void_p, c_int, c_int, c_int]),
    'cms_program_sync_count': (c_int, [c_void_p]),
    'cms_program_set_sync_flags': (c_int, [c_void_p, c_void_p, c_int]),
    'cms_program_add_aspp_gather': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, _P(c_int), _P(c_int), c_int, c_int,
                                            c_int, c_int, c_int, c_int, c_int, c_int]),
    'cms_program_add_aspp_spread': (c_int, [c_void_p, c_void_p, c_void_p, c_int, _P(c_int), _P(c_int), c_int, c_int, c_int,
                                            c_int, c_int, c_int, c_int, c_int]),
    'cms_program_add_bn': (c_int, [c_void_p, _P(BnOp), c_int, c_int]),
    'cms_program_size': (c_int, [c_void_p]),
    'cms_program_run': (c_int, [c_void_p, c_int, c_int, _P(c_void_p), c_int]),
    'cms_program_run_pair': (c_int, [c_void_p, _P(c_void_p), c_int, c_void_p, _P(c_void_p), c_int]),
    'cms_program_set_timing': (c_int, [c_void_p, c_int]),
    'cms_program_read_timing': (c_int, [c_void_p, _P(C.c_double), _P(C.c_double), _P(C.c_long), _P(C.c_double),
                                        _P(C.c_long)]),
    'cms_conv_igemm_f32': (c_int, [_P(ConvDesc), c_void_p]),
    'cms_conv_wgrad_f32': (c_int, [_P(WgradDesc), c_void_p]),
    'cms_conv_pack_transpose_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'cms_conv_pack_transpose_batch_f32': (c_int, [c_void_p, c_int, c_int, c_void_p]),
}

fn = {}
for _name, (_res, _args) in PROTOTYPES.items():
    fn[_name] = _proto(_name, _res, _args)


class CmsError(RuntimeError):
    pass


def check(rc, what=''):
    """Raise on a negative return code. CMS_EINVAL maps to ValueError (the reference raises ValueError for bad
    configuration), everything else to CmsError."""
    if rc == 0:
        return
    msg = fn['cms_last_error']().decode(errors='replace')
    if rc == -1:
        raise ValueError('{}: {}'.format(what or 'cutmixseg', msg))
    raise CmsError('{} failed (code {}): {}'.format(what or 'cutmixseg', rc, msg))


def version():
    return fn['cms_version']()
