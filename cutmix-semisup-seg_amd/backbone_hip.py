"""
Static executor of the DeepLab v2 body (layer1..layer4 + ASPP head) on the hand-written MFMA convolution kernels
(csrc/conv.hip) -- forward AND backward, replacing the autograd graph of ~450 library ops per pass that
`Bottleneck.forward` / `Classifier_Module.forward` of architectures/deeplab2.py:89-128 expand to.

Per bottleneck (frozen BatchNorm folded into scale/bias, deeplab2.py:92-107):

    forward      a   = relu(conv1(x)*s1 + b1)                 1x1 (stride on this conv, deeplab2.py:70)
                 b   = relu(conv2(a)*s2 + b2)                 3x3 dilated
                 out = relu(conv3(b)*s3 + b3 + res)           res = x  or  conv_d(x)*sd + bd
    backward     given dC = d loss / d (pre-ReLU sum), already masked by [out > 0]:
                 dW3 += wgrad(b, dC)*s3      dU2 = dgrad3(dC) * [b > 0]
                 dW2 += wgrad(a, dU2)*s2     dU1 = dgrad2(dU2) * [a > 0]
                 dWd += wgrad(x, dC)*sd      dres = dgrad_d(dC)           (or dres = dC for an identity shortcut)
                 dW1 += wgrad(x, dU1)*s1     dC_prev = (dgrad1(dU1) + dres) * [x > 0]
    Every ReLU mask, residual add, BN affine and bf16 rounding happens in a convolution epilogue; weight gradients
    are accumulated straight into the fp32 gradient arena. dgrad is the forward kernel on the transposed weights
    (BN scale folded, cms_conv_pack_transpose) with negated tap offsets.

With frozen BatchNorm (`--freeze_bn`; BN layers in eval mode) the statistics fold into scale / bias as above. Round 3:
BatchNorm on BATCH statistics (the reference CLI's default, train_seg_semisup_mask_mt.py:587; deeplab2.py:72-84) runs on
the executor too -- every unit becomes  u = conv(x);  y = relu(bn_batch(u) (+ res))  with the csrc/bn.hip launches recorded
into the same programs (cms_program_add_bn), see `_fwd_unit_bn` / `_bwd_unit_bn` below.
"""
import os
import torch

from . import ops
from .arena import ensure_arena


def executors_of(module):
    """Every MFMA executor built for `module` (one per compute dtype); empty for networks on the library engine."""
    exs = getattr(module, '_hip_executors', None)
    if exs:
        return list(exs.values())
    ex = getattr(module, '_hip_executor', None)
    return [] if ex is None else [ex]


class _Conv(object):
    __slots__ = ('wkey', 'bn', 'taps', 'ntaps', 'neg_taps', 'stride', 'cin', 'cout', 'scale', 'bias', 'wT', 'ksize',
                 'pad', 'dil', 'wdot', 'dbeta', 'wT_raw')

    def __init__(self, wkey, bn, conv):
        self.wkey, self.bn = wkey, bn
        kh, kw = conv.kernel_size
        self.ksize, self.pad, self.dil = kh, conv.padding[0], conv.dilation[0]
        self.taps = ops.conv_taps(kh, kw, conv.dilation[0], conv.padding[0])
        self.neg_taps = [(-dy, -dx) for dy, dx in self.taps]
        self.stride = conv.stride[0]
        self.cin, self.cout = conv.in_channels, conv.out_channels
        self.scale = self.bias = self.wT = None
        self.wT_raw = None                 # dgrad operand WITHOUT a folded BatchNorm scale (batch-statistics passes)
        self.wdot = self.dbeta = None      # side outputs of the weight gradient when the BN affine trains


class _Block(object):
    __slots__ = ('c1', 'c2', 'c3', 'cd')


class DeepLabHipExecutor(object):
    def __init__(self, net, dtype=torch.bfloat16):
        """`dtype`: torch.bfloat16 = the throughput configuration (bf16 activations / weights on the bf16 MFMA, fp32
        accumulation); torch.float32 = the PARITY configuration (fp32 activations, the fp32 master weights themselves as
        operands, f32-input MFMA -- csrc/conv_f32.hip)."""
        self._init_common(net, dtype)
        self.num_classes = net.num_classes
        self._add_blocks('', [getattr(net, 'layer{}'.format(li)) for li in range(1, 5)])
        # the stem (7x7/2 convolution + frozen BatchNorm + ReLU + ceil-mode max-pool) on csrc/stem.hip
        self.stem_wkey, self.stem_bn, self.stem_ceil = 'conv1.weight', 'bn1', True
        self.stem_w147 = None
        self._stem_version = -1
        head = net.layer5.conv2d_list
        self.aspp_keys = ['layer5.conv2d_list.0', 'layer5.conv2d_list.1']
        self.aspp_taps = []
        for i in range(2):
            self.aspp_taps += ops.conv_taps(3, 3, head[i].dilation[0], head[i].padding[0])
        self.aspp_neg_taps = [(-dy, -dx) for dy, dx in self.aspp_taps]
        dev = self.arena.device
        C = self.num_classes
        if C > 32:
            raise NotImplementedError('ASPP head kernel is specialised for <= 32 classes')
        # the head as ONE pass over the 2048-channel activation (csrc/aspp.hip): rows of Wall = tap * C + class over the
        # 18 taps of the two live dilations, padded to a multiple of 128 rows
        self.aspp_zc = (18 * C + 127) // 128 * 128
        self.aspp_wall = torch.zeros(1, self.aspp_zc, 2048, dtype=self.dtype, device=dev)     # forward / wgrad layout
        self.aspp_wallT = torch.zeros(1, 2048, self.aspp_zc, dtype=self.dtype, device=dev)    # dgrad operand
        self._aspp_version = -1
        self.aspp_bias = torch.zeros(C, dtype=torch.float32, device=dev)

    def _init_common(self, net, dtype=torch.bfloat16):
        if dtype not in (torch.bfloat16, torch.float32):
            raise TypeError('executor dtype must be torch.bfloat16 or torch.float32')
        self.net = net
        self.dtype = dtype
        self.trainable = any(p.requires_grad for p in net.parameters())
        # BatchNorm affine parameters that train (torchvision-style backbones) move with every optimizer step
        self.bn_trainable = any(p.requires_grad for m in net.modules() if 'BatchNorm' in type(m).__name__
                                for p in m.parameters())
        self.arena = ensure_arena(net, with_grad=self.trainable, with_bf16=(dtype == torch.bfloat16))
        self.blocks = []
        self._layer_first = []
        self._affine_ready = False
        self._bn_idx = None
        self._side = None
        self._pack_plan = None
        self.grad_hook = None      # callable(block_index): weight gradients of that bottleneck are enqueued
        self.data_grad_only = False   # backward computes d/d input only (VAT direction: torch.autograd.grad wrt eps)
        self.overlap_wgrad = True
        # number of extra streams the weight gradients of a bottleneck are spread over (independent launches). With the
        # 128 x 128 kernel (~1.5 workgroups per CU per launch: each launch fills the machine) one stream beside the
        # data-gradient chain was enough (profiles/r02f_wgrad_streams.txt: 1 -> 443-445 img/s, 2 -> 433, 3 -> 443). The
        # eight-phase 256 x 256 kernel (csrc/wgrad8.hip) puts ~56 workgroups = 56 CUs behind a launch: TWO streams of them
        # fill the ~124 CUs the data-gradient convolution (132 tiles, one per CU) leaves free (profiles/r04ag-ai_*: 1 stream
        # 561 img/s at its best split, 2 streams 575-601, 3 streams 500-563). CMS_WGRAD_STREAMS overrides (A/B switch).
        self.wgrad_streams = int(os.environ.get('CMS_WGRAD_STREAMS', '2' if os.environ.get('CMS_WGRAD8', '1') != '0' else '1'))
        # Weight gradients of this many consecutive bottlenecks go out as ONE grouped launch per kind (ops.conv_wgrad_group)
        # on the weight-gradient stream, behind the data-gradient chain of the stretch; 0 = one launch per layer (round 1-3).
        # CMS_WGRAD_GROUP sets it (A/B switch, read once).
        self.wgrad_group_blocks = int(os.environ.get('CMS_WGRAD_GROUP', '0'))
        # ReLU masks of the block outputs as bits (written by the expansion's epilogue, read by the data gradients that need the
        # activation only for its sign); CMS_RELU_BITS=0 switches them off (A/B)
        self.relu_bits = os.environ.get('CMS_RELU_BITS', '1') != '0'
        self.relu_bits_inner = os.environ.get('CMS_RELU_BITS_INNER', '1') != '0'     # ... of a1 / a2 as well (round 5; A/B)
        # The backward pass normally ends with the main stream waiting for the weight-gradient stream(s): whoever reads a
        # gradient afterwards finds it complete. A caller that knows where it next touches the gradients (the training step:
        # at the gradient exchange / optimizer) sets this and calls `join_wgrad()` there instead -- what follows the body's
        # backward on the main stream (max-pool and stem backward, 0.3 ms) then overlaps the tail of layer1's weight
        # gradients (0.37 ms of small launches, profiles/r04n_step_timeline.txt) instead of waiting for it.
        self.defer_wgrad_join = False
        self._pending_join = None
        self.head_wgrad_side = os.environ.get('CMS_HEAD_WGRAD_SIDE', '1') != '0'
        # EXPERIMENT (round 5, CMS_FWD_SPLIT bit 0: trainable network, bit 1: network without gradients): a recorded frozen-BN forward
        # pass issues every convolution as TWO launches over the halves of the batch on two streams (samples are independent; both
        # halves write slices of the same buffers, so the backward pass is unchanged): more, smaller launches in flight
        self.fwd_split = int(os.environ.get('CMS_FWD_SPLIT', '0'))
        self._sides = []
        self.conv_tile = 0         # experiment knob: force a tile shape on the 128-multiple layers (tools, bench)
        self.tile_rules = {}       # output channels -> tile code (per-layer choice against the workgroup-count staircase)
        self._wT_version = -1
        self.version = 0          # bumped whenever the weights change (optimizer step / load_state_dict)
        # Launch programs (ops.Program, csrc/program.hip): the host side of a pass is recorded once per input shape over
        # persistent buffers and replayed from C++ with one call -- False runs every launch eagerly from Python
        # (tools that probe single layers, the DeepLab v3+ executor below)
        self.use_programs = True
        self._programs = {}       # key -> forward program
        self._generation = 0      # forward passes issued; a backward pass checks that its activations are still there
        # what was enqueued through programs so far (bench.py): algorithmic MFMA FLOPs, body-convolution launches
        # and their algorithmic bytes, ASPP-head launches and bytes
        self.issued = dict(flops=0.0, conv_launches=0, conv_bytes=0.0, head_launches=0, head_bytes=0.0, head_bytes_alg=0.0, floor_s=0.0)
        net.register_load_state_dict_post_hook(lambda module, incompatible: self.invalidate())

    def _add_blocks(self, prefix, layers):
        """`layers`: the four nn.Sequential stages of bottlenecks; `prefix`: state-dict prefix of 'layer1' .. 'layer4'."""
        for li, layer in enumerate(layers):
            self._layer_first.append(len(self.blocks))
            for bi, blk in enumerate(layer):
                pre = '{}layer{}.{}'.format(prefix, li + 1, bi)
                b = _Block()
                b.c1 = _Conv(pre + '.conv1.weight', pre + '.bn1', blk.conv1)
                b.c2 = _Conv(pre + '.conv2.weight', pre + '.bn2', blk.conv2)
                b.c3 = _Conv(pre + '.conv3.weight', pre + '.bn3', blk.conv3)
                b.cd = None
                if blk.downsample is not None:
                    b.cd = _Conv(pre + '.downsample.0.weight', pre + '.downsample.1', blk.downsample[0])
                self.blocks.append(b)

    # ------------------------------------------------------------------------------------------ weights / affine
    def invalidate(self):
        """Call after the fp32 weights changed outside the fused optimizer (load_state_dict, manual edits)."""
        self.arena.refresh_bf16()
        self._affine_ready = False
        self.version += 1

    def weights_changed(self, bn_too=False):
        """The fp32 AND bf16 arenas were updated by an optimizer / EMA kernel. `bn_too`: BatchNorm state moved as well
        (the EMA runs over the running statistics and affine parameters too, optim_weight_ema.py:21-25; even for
        equal source and target its three roundings can move a value by an ulp) -> re-fold scale / bias."""
        self.version += 1
        if bn_too or self.bn_trainable:
            self._affine_ready = False

    def block_grad_offsets(self):
        """Arena offset of the first tensor of every bottleneck (state_dict order: conv1.weight comes first)."""
        return [self.arena.by_key[b.c1.wkey].offset for b in self.blocks]

    def layer_first_blocks(self):
        return list(self._layer_first)

    def _all_convs(self):
        for b in self.blocks:
            for c in (b.c1, b.c2, b.c3, b.cd):
                if c is not None:
                    yield c

    def _build_affine_tables(self):
        """Gather indices of every BatchNorm (weight, bias, running_mean, running_var) element in the arena, in the
        order of `_all_convs()`, so that folding all 104 layers is a handful of launches instead of ~600."""
        a = self.arena
        idx = {k: [] for k in ('weight', 'bias', 'running_mean', 'running_var')}
        off = 0
        spans = []
        for c in self._all_convs():
            for k in idx:
                s = a.by_key[c.bn + '.' + k]
                idx[k].append(torch.arange(s.offset, s.offset + s.count, dtype=torch.int64))
            spans.append((c, off, off + c.cout))
            off += c.cout
        stem_span = None
        if getattr(self, 'stem_bn', None) is not None:
            for k in idx:
                s = a.by_key[self.stem_bn + '.' + k]
                idx[k].append(torch.arange(s.offset, s.offset + s.count, dtype=torch.int64))
            stem_span = (off, off + a.by_key[self.stem_bn + '.weight'].count)
            off = stem_span[1]
        dev = a.device
        self._bn_idx = {k: torch.cat(v).to(dev) for k, v in idx.items()}
        self._scale_all = torch.zeros(off, dtype=torch.float32, device=dev)
        self._bias_all = torch.zeros(off, dtype=torch.float32, device=dev)
        for c, lo, hi in spans:
            c.scale, c.bias = self._scale_all[lo:hi], self._bias_all[lo:hi]
        if stem_span is not None:
            self.stem_scale, self.stem_bias = self._scale_all[stem_span[0]:stem_span[1]], self._bias_all[stem_span[0]:stem_span[1]]
        if self.bn_trainable and self.trainable:
            self._wdot_all = torch.zeros(off, dtype=torch.float32, device=dev)
            self._dbeta_all = torch.zeros(off, dtype=torch.float32, device=dev)
            for c, lo, hi in spans:
                c.wdot, c.dbeta = self._wdot_all[lo:hi], self._dbeta_all[lo:hi]

    def _refresh_affine(self):
        """scale = gamma / sqrt(var + eps), bias = beta - mean * scale  (frozen BN, deeplab2.py:92-107)."""
        if self._bn_idx is None:
            self._build_affine_tables()
        flat, ix = self.arena.flat, self._bn_idx
        # (round 5) ONE launch instead of nine tensor ops per refresh (the teacher's statistics move with every EMA step)
        ops.bn_fold(flat, ix['weight'], ix['bias'], ix['running_mean'], ix['running_var'], 1e-5, self._scale_all, self._bias_all)
        self._affine_ready = True

    def _wbuf(self):
        """The flat buffer the convolution operands are views of: the bf16 copy the fused optimizer maintains, or -- in
        the fp32 parity configuration -- the fp32 master arena itself."""
        return self.arena.bf16 if self.dtype == torch.bfloat16 else self.arena.flat

    def _w(self, c):
        return self.arena.packed(c.wkey, self._wbuf())

    def _refresh_aspp_fwd(self):
        """Padded operands of the head (class axis -> 32 / 64 rows, the two live dilations stacked to 18 taps). Only
        when the weights moved since the last refresh (optimizer / EMA step, load_state_dict)."""
        if self._aspp_version == self.version:
            return
        C = self.num_classes
        a = self.arena
        for i, k in enumerate(self.aspp_keys):
            wk = a.packed(k + '.weight', self._wbuf())                       # (9, C, 2048): rows tap * C + class
            self.aspp_wall[0, 9 * C * i:9 * C * (i + 1)].copy_(wk.reshape(9 * C, 2048))
        torch.add(a.view(self.aspp_keys[0] + '.bias'), a.view(self.aspp_keys[1] + '.bias'), out=self.aspp_bias)
        self._aspp_version = self.version

    def _refresh_backward_weights(self):
        """dgrad operands wT[tap][ci][co] = bf16(w * scale[co]) of all 104 convolutions + the head: one launch."""
        if not self._affine_ready:
            self._refresh_affine()
        if self._pack_plan is None:
            triples = []
            for c in self._all_convs():
                w = self._w(c)
                c.wT = torch.empty((w.shape[0], w.shape[2], w.shape[1]), dtype=self.dtype, device=w.device)
                triples.append((w, c.wT, c.scale))
            if hasattr(self, 'aspp_wall'):
                triples.append((self.aspp_wall, self.aspp_wallT, None))
            self._pack_plan = ops.PackTransposePlan(triples)
        self._pack_plan.run()

    # ------------------------------------------------------------------------------------------ forward
    @staticmethod
    def _out_hw(h, w, stride):
        return (h - 1) // stride + 1, (w - 1) // stride + 1

    def _fwd(self, x, c, relu, res=None, bits=False, halves=None):
        """`bits`: also write the ReLU mask of the output as bits (cms_conv_desc.mask_bits_out) and hang them on the returned
        tensor (`_cms_relu_bits`): the data gradient that needs this activation only for its sign reads 1/16 of the bytes.
        `halves`: two streams -- the launch goes out as two, over the two halves of the batch (see `fwd_split`)."""
        n, h, w, _ = x.shape
        ho, wo = self._out_hw(h, w, c.stride)
        mb = None
        if bits and relu and self.relu_bits and self.dtype == torch.bfloat16 and c.cout % 32 == 0:
            mb = torch.empty((n, ho, wo, c.cout // 8), dtype=torch.uint8, device=x.device)
        if halves is not None:
            y = torch.empty((n, ho, wo, c.cout), dtype=x.dtype, device=x.device)
            for hi, (s0, s1) in enumerate(((0, n // 2), (n // 2, n))):
                with torch.cuda.stream(halves[hi]):
                    ops.conv_igemm(x[s0:s1], self._w(c), c.taps, stride=c.stride, out_hw=(ho, wo), scale=c.scale, bias=c.bias,
                                   res=None if res is None else res[s0:s1], relu=relu, tile=self._tile(c.cout),
                                   mask_bits_out=None if mb is None else mb[s0:s1], out=y[s0:s1])
            if mb is not None:
                y._cms_relu_bits = mb
            return y
        y = ops.conv_igemm(x, self._w(c), c.taps, stride=c.stride, out_hw=(ho, wo), scale=c.scale, bias=c.bias,
                           res=res, relu=relu, tile=self._tile(c.cout), mask_bits_out=mb)
        if mb is not None:
            y._cms_relu_bits = mb
        return y

    def _tile(self, cout):
        """Workgroup tile code for a convolution with `cout` output channels (0 = the library's choice)."""
        if cout in self.tile_rules:
            return self.tile_rules[cout]
        return self.conv_tile if (self.conv_tile and cout % 128 == 0) else 0

    def fwd_begin(self, x, save):
        """x: bf16 NHWC (N, h, w, 64) = stem + max-pool output. The forward pass in three pieces (begin / one call per
        bottleneck / end) so that a caller can interleave the launches of two networks on two streams (`_BodyPairFn`)."""
        bn = self.batch_statistics()
        if not bn and not self._affine_ready:
            self._refresh_affine()
        groups = self.bn_groups() if bn else 1
        if x.shape[0] % groups != 0:
            raise ValueError('{} sample groups do not divide a batch of {}'.format(groups, x.shape[0]))
        return {'cur': x, 'saved': [] if save else None, 'bn': bn, 'groups': groups}

    def fwd_block(self, st, bi):
        if st.get('bn'):
            return self._fwd_block_bn(st, bi)
        b, cur = self.blocks[bi], st['cur']
        # (round 5) a1 / a2 get mask bits too: both kernels of cms_conv_igemm write and read them (the library alone decides
        # which one runs a launch), so the data gradients of conv2 / conv3 no longer re-read these activations for their sign
        inner = st['saved'] is not None and self.relu_bits_inner
        hv = st.get('halves')
        a1 = self._fwd(cur, b.c1, True, bits=inner, halves=hv)
        a2 = self._fwd(a1, b.c2, True, bits=inner, halves=hv)
        res = cur if b.cd is None else self._fwd(cur, b.cd, False, halves=hv)
        st['cur'] = self._fwd(a2, b.c3, True, res=res, bits=st['saved'] is not None, halves=hv)
        if st['saved'] is not None:
            st['saved'].append((cur, a1, a2))

    def fwd_end(self, st):
        """-> (logits fp32 NCHW, saved)"""
        cur, saved = st['cur'], st['saved']
        self._refresh_aspp_fwd()
        n, h, w, _ = cur.shape
        # Z[n][tap*C + c] = <W[tap][c], x> as ONE 1x1 GEMM (the activation is read once, not once per tap), then the 18
        # shifted planes of every class are summed (csrc/aspp.hip)
        z = torch.empty((n, self.aspp_zc, h, w), dtype=torch.float32, device=cur.device)
        ops.conv_igemm(cur, self.aspp_wall, [(0, 0)], out_f32_nchw=z, cout_real=self.aspp_zc)
        logits = ops.aspp_gather_fwd(z, self.aspp_bias, self.aspp_taps, self.num_classes)
        if saved is not None:
            saved.append(cur)
        return logits, saved

    # ------------------------------------------------------------------------------------------ batch-statistics passes
    def batch_statistics(self):
        """True when the network's BatchNorm layers are in training mode (no --freeze_bn): the passes normalise with batch
        statistics and move the running statistics (momentum), in the student and in the train-mode teacher alike (Q3, Q4)."""
        return any(self._bn_module(c).training for c in self._all_convs())      # (the executor's own layers: DeepLab v3+ keeps
                                                                                # its HEAD on batch statistics, not the backbone)

    def bn_groups(self):
        """Sample groups of the next batch-statistics pass (architectures/deeplab2.py:set_sample_groups): the batch is that
        many equal runs of samples, each normalised with its own statistics."""
        fn = getattr(self.net, 'sample_groups', None)
        return 1 if fn is None else int(fn())

    def _dist_group(self):
        """Process group of the SyncBN exchanges: the one the training step hands to the network (`net.dist_group`, set by
        CutMixMeanTeacherStep: the group its gradient exchange uses) -- None = the default group."""
        return getattr(self.net, 'dist_group', None)

    def _bn_module(self, c):
        mods = self.__dict__.setdefault('_bn_modules', {})
        m = mods.get(c.bn)
        if m is None:
            m = mods[c.bn] = self.net.get_submodule(c.bn)
        return m

    def _fwd_unit_bn(self, x, c, relu, res=None, save=True, groups=1):
        """y = relu(batch_norm(conv(x)) (+ res)) as THREE launches on persistent buffers: raw convolution (round 5: its epilogue
        leaves per-tile channel sums), statistics (round 5: adds the tile sums and finalises -- scale / shift, running statistics,
        batch counter; under data parallelism or in fp32: the atomics-free reduction over u of rounds 3-4), normalise + residual +
        ReLU (+ the ReLU mask as bits). -> (y, saved), saved = (u, mask bits or y or None, mean, rstd, backward sums, workspace,
        groups) for `_bwd_unit_bn`. (/root/reference/architectures/deeplab2.py:72-84 in training mode.)"""
        n, h, w, _ = x.shape
        ho, wo = self._out_hw(h, w, c.stride)
        G = int(groups)
        grp = self._dist_group()
        world = ops._world(grp)
        # (round 5) single process, bf16: the convolution's epilogue leaves per-tile channel sums of what it stores and the
        # statistics launch only adds those up -- the pass over u it used to be is gone (cms_conv_desc.stats_out)
        # (round 6) under data parallelism too: the tile sums become the per-group sums the ranks all-reduce (below)
        st = {'groups': G} if (self.dtype == torch.bfloat16 and _fused_bn_stats()) else None
        u = ops.conv_igemm(x, self._w(c), c.taps, stride=c.stride, out_hw=(ho, wo), tile=self._tile(c.cout), stats=st)
        a, bn, C = self.arena, self._bn_module(c), c.cout
        npix = n * ho * wo
        dev = x.device
        bsums = torch.empty(G * 2 * C, dtype=torch.float64, device=dev) if save else None
        mean, rstd, scale, shift = (torch.empty(G * C, dtype=torch.float32, device=dev) for _ in range(4))
        ws = ops.bn_workspace(npix, C, dev, G)      # this unit's: tile counters + partial sums (forward, then backward)
        if st is not None and st['tile_rows'] > 0 and world == 1:
            ops.bn_op('finalize_tiles', c=C, dtype=self.dtype, n_pixels=npix, groups=G, eps=bn.eps, momentum=bn.momentum,
                      tile_rows=st['tile_rows'], ws=st['tile_sums'], gamma=a.view(c.bn + '.weight'), beta=a.view(c.bn + '.bias'),
                      mean=mean, rstd=rstd, scale=scale, shift=shift, running_mean=a.view(c.bn + '.running_mean'),
                      running_var=a.view(c.bn + '.running_var'), counter=bn.num_batches_tracked)
        elif world > 1:
            # SyncBN on the executor (round 4; SURVEY 8(e) "BN statistics"): per-group (sum x, sum x^2) -> ONE all-reduce of
            # [G][2][C] doubles between two launches of the recorded pass (a host op of the program) -> the groups finalised in
            # order with the pixel count of ALL ranks. The fused single-process launch ('stats') does the same without the exchange.
            fsums = torch.empty(G * 2 * C, dtype=torch.float64, device=dev)
            if st is not None and st['tile_rows'] > 0:
                # (round 6) the convolution's epilogue left per-tile (sum x, sum x^2): added up per group in a fixed order (the
                # finalising kernel with a `sums` output) instead of a second pass over u
                ops.bn_op('sums_tiles', c=C, dtype=self.dtype, n_pixels=npix, groups=G, tile_rows=st['tile_rows'],
                          ws=st['tile_sums'], sums=fsums)
            else:
                ops.bn_op('reduce', c=C, dtype=self.dtype, n_pixels=npix, groups=G, x=u, sums=fsums, ws=ws)
            ops.host_call(lambda t=fsums, g_=grp: ops._allreduce_sum(t, g_))
            for g in range(G):
                sl = slice(g * C, (g + 1) * C)
                ops.bn_op('finalize', c=C, count=float(npix // G) * world, eps=bn.eps, momentum=bn.momentum,
                          sums=fsums[g * 2 * C:(g + 1) * 2 * C], gamma=a.view(c.bn + '.weight'), beta=a.view(c.bn + '.bias'),
                          mean=mean[sl], rstd=rstd[sl], scale=scale[sl], shift=shift[sl],
                          running_mean=a.view(c.bn + '.running_mean'), running_var=a.view(c.bn + '.running_var'),
                          counter=bn.num_batches_tracked)
        else:
            ops.bn_op('stats', c=C, dtype=self.dtype, n_pixels=npix, groups=G, eps=bn.eps, momentum=bn.momentum, x=u, ws=ws,
                      gamma=a.view(c.bn + '.weight'), beta=a.view(c.bn + '.bias'), mean=mean, rstd=rstd, scale=scale, shift=shift,
                      running_mean=a.view(c.bn + '.running_mean'), running_var=a.view(c.bn + '.running_var'),
                      counter=bn.num_batches_tracked)
        y = torch.empty_like(u)
        # (round 5) the ReLU mask of the backward passes as BITS beside y: they then stream u, dy and 1/16 of a tensor instead of y
        bits = torch.empty(npix * C // 8, dtype=torch.uint8, device=dev) if (relu and save and _bn_mask_bits()) else None
        ops.bn_op('apply', c=C, dtype=self.dtype, n_pixels=npix, groups=G, relu=relu, x=u, res=res, y=y, scale=scale,
                  shift=shift, mask_bits=bits)
        return y, (u, (y if bits is None else bits) if relu else None, mean, rstd, bsums, ws, G)

    def _bstats(self, s, unit):
        """Descriptor of a unit's backward statistics for the data-gradient launch that writes the gradient of its output
        (`ops.conv_igemm(..., mode=1, stats=...)`, cms_conv_desc.bstats_*), or None: `s` = the unit's saved tuple, `unit` = 1 / 2 / 3.
        CMS_BN_BWD_STATS (read per recording): 0 = off, 3 = the wide unit 3 only (default: 329.8 -> 335.1 img/s; every unit: 333 -- for the
        narrow units the finalising launch costs what the reduction it replaces did, profiles/r05u_*), 1 = every unit."""
        mode = os.environ.get('CMS_BN_BWD_STATS', '3')
        # (round 6: under data parallelism too -- xhat uses the unit's GLOBAL mean / rstd, the per-tile sums are local and the
        # per-group sums they add up to are all-reduced like the reduction kernel's, `_bwd_unit_bn`)
        if mode == '0' or (mode == '3' and unit != 3) or self.dtype != torch.bfloat16:
            return None
        u, yb, mean, rstd, _sums, _ws, G = s
        if yb is not None and yb.dtype != torch.uint8:
            return None                         # the mask is the stored activation (CMS_BN_MASK_BITS=0): the reduction kernel reads it
        return {'groups': G, 'u': u, 'mean': mean, 'rstd': rstd, 'bits': yb}

    def _bwd_unit_bn(self, dy, saved, c, want_res, dy_bits=None, tile_stats=None):
        """Backward of the normalisation of one unit: dy (gradient wrt y) -> (du = gradient wrt the convolution output,
        dres = gradient wrt the residual input or None). The ReLU mask comes from the stored y."""
        u, y, mean, rstd, sums, ws, G = saved   # `sums` is overwritten by the reduction; `ws`: the unit's workspace
        C = c.cout
        npix = u.numel() // C
        bits = None
        if y is not None and y.dtype == torch.uint8:        # the mask as bits (`_fwd_unit_bn`)
            bits, y = y, None
        if dy_bits is not None:
            # `dy` arrives UNMASKED with the mask of the tensor it is the gradient of (the downsample unit behind a block output
            # whose masked gradient `dres` is no longer materialised): this unit has no ReLU of its own, the bits take its place
            assert y is None and bits is None
            bits = dy_bits
        if tile_stats is not None and tile_stats.get('tile_rows', 0) > 0:
            # (round 5) the launch that wrote dy left per-tile (sum d, sum d xhat): no pass over u, dy and the mask
            assert dy_bits is None
            ops.bn_op('sums_tiles', c=C, dtype=self.dtype, n_pixels=npix, groups=G, tile_rows=tile_stats['tile_rows'],
                      ws=tile_stats['tile_sums'], sums=sums)
        else:
            ops.bn_op('reduce_bwd', c=C, dtype=self.dtype, n_pixels=npix, groups=G, x=u, dy=dy, y=y, mean=mean, rstd=rstd, sums=sums,
                      ws=ws, mask_bits=bits)
        grp = self._dist_group()
        world = ops._world(grp)
        if world > 1:                        # SyncBN: (sum dy', sum dy' xhat) of every group over all ranks
            ops.host_call(lambda t=sums, g_=grp: ops._allreduce_sum(t, g_))
        du = torch.empty_like(u)
        dres = torch.empty_like(u) if want_res else None
        ops.bn_op('bwd_apply', c=C, dtype=self.dtype, n_pixels=npix, groups=G, count=(npix // G) * world, x=u, dy=dy, y=y, dx=du, dres=dres,
                  mask_bits=bits, mean=mean,
                  rstd=rstd, gamma=self.arena.view(c.bn + '.weight'), sums=sums)
        return du, dres

    def _fwd_block_bn(self, st, bi):
        b, cur = self.blocks[bi], st['cur']
        save, G = st['saved'] is not None, st.get('groups', 1)
        a1, s1 = self._fwd_unit_bn(cur, b.c1, True, save=save, groups=G)
        a2, s2 = self._fwd_unit_bn(a1, b.c2, True, save=save, groups=G)
        sd = None
        if b.cd is None:
            res = cur
        else:
            res, sd = self._fwd_unit_bn(cur, b.cd, False, save=save, groups=G)
        out, s3 = self._fwd_unit_bn(a2, b.c3, True, res=res, save=save, groups=G)
        st['cur'] = out
        if st['saved'] is not None:
            st['saved'].append((cur, a1, a2, s1, s2, s3, sd))

    def _refresh_backward_weights_bn(self):
        """dgrad operands wT[tap][ci][co] = w^T (NO BatchNorm scale: the normalisation has its own backward) of all body
        convolutions + the head: one launch."""
        if self.__dict__.get('_pack_plan_bn') is None:
            triples = []
            for c in self._all_convs():
                w = self._w(c)
                c.wT_raw = torch.empty((w.shape[0], w.shape[2], w.shape[1]), dtype=self.dtype, device=w.device)
                triples.append((w, c.wT_raw, None))
            triples.append((self.aspp_wall, self.aspp_wallT, None))
            self._pack_plan_bn = ops.PackTransposePlan(triples)
        self._pack_plan_bn.run()

    def _dgrad_raw(self, du, c, res=None, in_hw=None, res_bits=None, bstats=None):
        """Data gradient of one convolution (+ res). `res_bits`: res is the UNMASKED gradient of a ReLU output and the bits are that
        ReLU's mask -- added as (bit ? res : 0) in the epilogue (cms_conv_desc.mask_gates_res)."""
        n, ho, wo, _ = du.shape
        if c.stride == 1:
            return ops.conv_igemm(du, c.wT_raw, c.neg_taps, res=res, mode=1, tile=self._tile(c.cin), mask_bits=res_bits,
                                  mask_gates_res=res_bits is not None, stats=bstats)
        assert res_bits is None and bstats is None
        return ops.conv_igemm(du, c.wT_raw, c.neg_taps, res=res, mode=1, out_hw=(ho, wo), out_stride=c.stride,
                              out_full_hw=in_hw, tile=self._tile(c.cin))

    def _wgrad_raw(self, du, x, c):
        ops.conv_wgrad(du, x, c.taps, self.arena.packed(c.wkey, self.arena.grad), stride=c.stride, wg_target=self._wg_target())

    def _backward_chain_bn(self, saved, dlg, want_w, sides, hook, box=None):
        """The launches of the backward pass with BatchNorm on batch statistics (recordable): head, then per bottleneck
        bn3 -> conv3 -> bn2 -> conv2 -> bn1 -> conv1 (+ the downsample branch). -> (dx, dwall or None)"""
        x4 = saved[-1]
        main = torch.cuda.current_stream()
        d = ops.aspp_spread_bwd(dlg, self.aspp_taps, self.aspp_zc, self.dtype)
        dwall = None
        if want_w:
            dwall = torch.empty((1, self.aspp_zc, 2048), dtype=torch.float32, device=d.device)
            ops.memset_zero(dwall)
            ops.conv_wgrad(d, x4, [(0, 0)], dwall)
            if box is not None:
                box['dwall'] = dwall
        # gradient wrt the block OUTPUT: bn3 applies its mask. `pend`: the backward statistics of that unit 3, left by the launch
        # that writes the gradient (`_bstats`)
        pend = self._bstats(saved[len(self.blocks) - 1][5], 3)
        dOut = ops.conv_igemm(d, self.aspp_wallT, [(0, 0)], mode=1, stats=pend)
        keep = []
        closes = set(self.bucket_starts())
        for bi in range(len(self.blocks) - 1, -1, -1):
            b = self.blocks[bi]
            xin, a1, a2, s1, s2, s3, sd = saved[bi]
            in_hw = (xin.shape[1], xin.shape[2])
            # (round 5) with the block output's ReLU mask as bits, its masked gradient `dres` (the widest tensor of the block) is
            # never written: the identity shortcut adds (bit ? dOut : 0) in conv1's data-gradient epilogue, the downsample unit's
            # backward passes read dOut through the same bits
            # (the gated shortcut `mask_gates_res` exists in the bf16 data-gradient kernels only: an fp32 identity block keeps `dres`)
            bits3 = s3[1] if (s3[1] is not None and s3[1].dtype == torch.uint8 and _bn_gate_shortcut()
                              and (b.cd is not None or (b.c1.stride == 1 and self.dtype == torch.bfloat16))) else None
            du3, dres = self._bwd_unit_bn(dOut, s3, b.c3, bits3 is None, tile_stats=pend)
            st2 = self._bstats(s2, 2)
            da2 = self._dgrad_raw(du3, b.c3, bstats=st2)
            du2, _ = self._bwd_unit_bn(da2, s2, b.c2, False, tile_stats=st2)
            st1 = self._bstats(s1, 1)
            da1 = self._dgrad_raw(du2, b.c2, bstats=st1)
            du1, _ = self._bwd_unit_bn(da1, s1, b.c1, False, tile_stats=st1)
            dud = None
            if b.cd is not None:
                dud, _ = self._bwd_unit_bn(dres if bits3 is None else dOut, sd, b.cd, False, dy_bits=bits3)
            if want_w:
                jobs = [(du3, a2, b.c3), (du2, a1, b.c2)] + ([(dud, xin, b.cd)] if b.cd is not None else []) + [(du1, xin, b.c1)]
                if sides:
                    for sd_ in sides:
                        ops.stream_wait(sd_, main)
                    keep.append((du3, du2, du1, dud))
                    for ji, (du_, x_, c_) in enumerate(jobs):
                        with torch.cuda.stream(sides[ji % len(sides)]):
                            self._wgrad_raw(du_, x_, c_)
                    if len(sides) > 1 and bi in closes:
                        for sd_ in sides[1:]:
                            ops.stream_wait(sides[0], sd_)
                    with torch.cuda.stream(sides[0]):
                        hook(bi)
                else:
                    for du_, x_, c_ in jobs:
                        self._wgrad_raw(du_, x_, c_)
                    hook(bi)
            # the launch that completes the gradient wrt this block's input = the previous block's output also takes that
            # block's unit-3 backward statistics
            pend = self._bstats(saved[bi - 1][5], 3) if (bi > 0 and b.c1.stride == 1) else None
            if b.cd is None and bits3 is not None:
                dOut = self._dgrad_raw(du1, b.c1, res=dOut, in_hw=in_hw, res_bits=bits3, bstats=pend)
            else:
                dx = dres if b.cd is None else self._dgrad_raw(dud, b.cd, in_hw=in_hw)
                dOut = self._dgrad_raw(du1, b.c1, res=dx, in_hw=in_hw, bstats=pend)
        if not self.defer_wgrad_join:
            for sd_ in sides:
                ops.stream_wait(main, sd_)
        self._retire_keep(keep)
        return dOut, dwall

    # ------------------------------------------------------------------------------------------ stem
    def _stem_prepare(self):
        if not self._affine_ready:
            self._refresh_affine()
        if self._stem_version != self.version:
            self.stem_w147 = ops.stem_pack_weights(self.arena.packed(self.stem_wkey, self._wbuf()), out=self.stem_w147)
            self._stem_version = self.version

    def stem(self, x):
        """(N, 3, H, W) image batch -> NHWC (N, hp, wp, 64) input of the body: 7x7/2 convolution + frozen BatchNorm +
        ReLU + max-pool (deeplab2.py:183-186) on csrc/stem.hip, with its own backward pass (weight gradient into the
        arena; image gradient only when the input asks for one: VAT)."""
        w = self.__dict__.get('_stem_param')
        if w is None:                     # (a walk over all named parameters: once, not per pass)
            w = self.__dict__['_stem_param'] = dict(self.net.named_parameters())[self.stem_wkey]
        # `save` of the body pass this stem feeds (run_body / run_body_pair decide the same way): its recorded program's input buffer
        # is where the pooled map is written (CMS_STEM_DIRECT=0: a new tensor + a copy, as in rounds 2-5)
        save = (torch.is_grad_enabled() and (self.trainable or x.requires_grad)) if (self.use_programs and _stem_direct()) else None
        return _StemFn.apply(x.contiguous(), w, self, save)

    def _stem_destination(self, shape, save):
        """Input buffer of the CACHED forward program for a body input of `shape` (N, h, w, 64), or None (not recorded yet, or a
        pass the executor runs launch by launch)."""
        key = ('fwd', tuple(int(v) for v in shape), bool(save), self._tile_key(), self._bn_key())
        prog = self._programs.get(key)
        return None if prog is None else prog.x_in

    def _prepare_forward(self):
        """Operand tables a forward pass reads (torch ops on the current stream, only when stale)."""
        if not self._affine_ready and not self.batch_statistics():
            self._refresh_affine()
        self._refresh_aspp_fwd()

    def _bn_key(self):
        """0 = frozen statistics (folded affine), else the number of sample groups of a batch-statistics pass."""
        return self.bn_groups() if self.batch_statistics() else 0

    def _tile_key(self):
        return (self.conv_tile, tuple(sorted(self.tile_rules.items())))

    MAX_PROGRAMS = 8

    def _program_lookup(self, key):
        """Cached program for `key`, marked most recently used -- or None after making room for a new recording (the
        LEAST recently used program goes; its buffers may still be in flight, hence the synchronisation)."""
        prog = self._programs.get(key)
        if prog is not None:
            self._programs[key] = self._programs.pop(key)          # dict order = recency
            return prog
        if len(self._programs) >= self.MAX_PROGRAMS:
            torch.cuda.synchronize()
            self._programs.pop(next(iter(self._programs)))
        return None

    def _eager_this_time(self, kind, shape, save):
        """Passes that keep activations for a backward pass (training shapes: they repeat every iteration) are recorded at
        once. A pass WITHOUT gradients is recorded only when its shape comes back: evaluating variable-sized images (the
        reference's Pascal validation loop, train_seg_semisup_mask_mt.py:484-517) then runs launch by launch instead of
        synchronising, evicting and re-recording on almost every image, and pins no activation sets in HBM."""
        if save:
            return False
        key = (kind, tuple(int(v) for v in shape), False, self._tile_key()) + ((self._bn_key(),) if kind == 'fwd' else ())
        if key in self._programs:
            return False
        seen = self.__dict__.setdefault('_seen_shapes', {})
        n = seen.get(key, 0) + 1
        if len(seen) > 256:
            seen.clear()
        seen[key] = n
        return n < 2

    def forward_program(self, shape, save):
        """The recorded forward pass for an input of `shape` (N, h, w, 64) -- recorded on first use, on the CURRENT
        stream (its stream 0). Attributes: x_in (persistent input buffer), logits, saved."""
        key = ('fwd', tuple(int(v) for v in shape), bool(save), self._tile_key(), self._bn_key())
        prog = self._program_lookup(key)
        if prog is None:
            self._prepare_forward()
            prog = ops.Program()
            x_in = torch.empty(tuple(shape), dtype=self.dtype, device=self.arena.device)
            main = torch.cuda.current_stream()
            split = (self.fwd_split >> (0 if self.trainable else 1)) & 1
            extra = [self._fwd_side_stream()] if (split and not key[-1] and int(shape[0]) >= 2 and int(shape[0]) % 2 == 0) else []
            with ops.recording(prog, [main] + extra):
                st = self.fwd_begin(x_in, save)
                if extra:
                    st['halves'] = [main, extra[0]]
                    ops.stream_wait(extra[0], main)          # the input buffer (and whatever the caller enqueued before the pass)
                for bi in range(len(self.blocks)):
                    prog.group = bi
                    self.fwd_block(st, bi)
                prog.group = len(self.blocks)
                if extra:
                    ops.stream_wait(main, extra[0])          # the head runs over the whole batch on the first stream
                logits, saved = self.fwd_end(st)
            prog.extra_streams = extra
            prog.x_in, prog.logits, prog.saved = x_in, logits, saved
            prog.bn = bool(key[-1])
            prog.bwd = {}
            prog.generation = -1
            self._programs[key] = prog
        return prog

    def programs(self):
        """Every recorded program of this executor (forward passes and their backward passes)."""
        out = []
        for p in self._programs.values():
            out.append(p)
            out += list(p.bwd.values())
        return out

    def bn_stat_sources(self):
        """BatchNorm launches of every recorded program by kind: which route the statistics of the batch-statistics passes took
        ('finalize_tiles' / 'sums_tiles' = from the convolution epilogues' tile sums; 'stats' / 'reduce' / 'reduce_bwd' = passes over
        the activations)."""
        out = {}
        for p in self.programs():
            for k, v in getattr(p, 'bn_kinds', {}).items():
                out[k] = out.get(k, 0) + v
        return out

    def _stamp(self, prog):
        self._generation += 1
        prog.generation = self._generation
        self._account(prog)

    def _account(self, prog):
        i = self.issued
        i['flops'] += prog.flops
        i['conv_launches'] += prog.conv_launches
        i['conv_bytes'] += prog.conv_bytes
        i['head_launches'] += prog.head_launches
        i['head_bytes'] += prog.head_bytes
        i['head_bytes_alg'] += prog.head_bytes_alg
        i['floor_s'] += prog.floor_s
        br = i.setdefault('by_route', {})
        for k, v in prog.by_route.items():
            r = br.setdefault(k, [0, 0.0, 0.0])
            r[0] += v[0]; r[1] += v[1]; r[2] += v[2]

    def forward(self, x, save):
        """-> (logits fp32 NCHW, token for `backward`)."""
        if not self.use_programs or self._eager_this_time('fwd', x.shape, save):
            st = self.fwd_begin(x, save)
            for bi in range(len(self.blocks)):
                self.fwd_block(st, bi)
            return self.fwd_end(st)
        prog = self.forward_program(x.shape, save)
        self._prepare_forward()
        if x.data_ptr() != prog.x_in.data_ptr():
            prog.x_in.copy_(x)
        prog.run([torch.cuda.current_stream()] + list(getattr(prog, 'extra_streams', [])))
        self._stamp(prog)
        # the logits buffer belongs to the program (the next pass of this shape overwrites it): hand out a copy
        return prog.logits.clone(), ((prog, prog.generation) if save else None)

    # ------------------------------------------------------------------------------------------ backward
    def _wgrad(self, du, x, c):
        if c.wdot is not None and self.dtype == torch.float32:
            # parity configuration: G = unscaled weight gradient from the f32 kernel, then <W, G> and sum_p dU as tensor ops
            g = torch.zeros_like(self._w(c))
            ops.conv_wgrad(du, x, c.taps, g, stride=c.stride)
            self.arena.packed(c.wkey, self.arena.grad).add_(g * c.scale.view(1, -1, 1))
            c.wdot.add_((g * self._w(c)).sum(dim=(0, 2)))
            c.dbeta.add_(du.sum(dim=(0, 1, 2)))
        elif c.wdot is not None and self._wfinish_takes(du, x, c):
            # (round 6) the wide layers of a backbone whose BatchNorm affine trains (DeepLab v3+) on the eight-phase kernel: the
            # UNSCALED gradient into a scratch tensor, d(beta) by a column sum of dU; `_wfinish_flush` (end of the pass) adds
            # scale * G to the arena and takes <W, G> (csrc/wfinish.hip)
            ops.conv_wgrad(du, x, c.taps, self.arena.packed(c.wkey, self._wscratch()), stride=c.stride, scale=None,
                           wg_target=self._wg_target())
            ops.channel_sum(du, c.dbeta)
            if all(c is not p for p in self._wfinish_pending):
                self._wfinish_pending.append(c)
        elif c.wdot is not None:
            ops.conv_wgrad(du, x, c.taps, self.arena.packed(c.wkey, self.arena.grad), stride=c.stride, scale=c.scale,
                           w_bf16=self._w(c), wdot=c.wdot, dbeta=c.dbeta)
        else:
            ops.conv_wgrad(du, x, c.taps, self.arena.packed(c.wkey, self.arena.grad), stride=c.stride, scale=c.scale,
                           wg_target=self._wg_target())

    # ---- trainable BatchNorm affine on the eight-phase weight-gradient kernel (round 6)
    def _wscratch(self):
        t = self.__dict__.get('_wscratch_buf')
        if t is None:
            t = self.__dict__['_wscratch_buf'] = torch.zeros_like(self.arena.grad)     # cleared again by every finishing launch
        return t

    @property
    def _wfinish_pending(self):
        return self.__dict__.setdefault('_wfinish_list', [])

    def _wfinish_takes(self, du, x, c):
        """bf16, atomics mode, and the launch WITHOUT side outputs would run on csrc/wgrad8.hip (Cout, Cin multiples of 256, enough
        pixels). CMS_V3_WGRAD8=0 keeps the side-output kernel everywhere (A/B)."""
        if self.dtype != torch.bfloat16 or ops.deterministic_wgrad() or os.environ.get('CMS_V3_WGRAD8', '1') == '0':
            return False
        key = (c.wkey, tuple(du.shape), tuple(x.shape))
        hit = self.__dict__.setdefault('_wfinish_ok', {}).get(key)
        if hit is None:
            hit = self._wfinish_ok[key] = bool(ops.conv_wgrad(du, x, c.taps, self.arena.packed(c.wkey, self._wscratch()),
                                                              stride=c.stride, scale=None, wg_target=self._wg_target(),
                                                              query_kernel=True))
        return hit

    def _wfinish_flush(self):
        """ONE finishing launch for the weight gradients `_wgrad` sent to the scratch tensor since the last flush -- on the
        current stream, which must be ordered behind all of them."""
        pend = self._wfinish_pending
        if not pend:
            return
        a = self.arena
        ops.wgrad_finish([(a.packed(c.wkey, self._wscratch()), a.packed(c.wkey, a.grad), self._w(c), c.scale, c.wdot) for c in pend])
        del pend[:]

    def _wg_target(self):
        """CUs a weight-gradient launch of the backward pass aims at (cms_wgrad_desc.wg_target): the launches run on
        `wgrad_streams` side streams beside the data-gradient chain, whose eight-phase convolutions hold 132 CUs."""
        if not self.overlap_wgrad:
            return 0
        t = os.environ.get('CMS_WG_TARGET')             # experiment: CUs a weight-gradient launch aims at
        if t:
            return int(t)
        return 56 if self.wgrad_streams >= 2 else 112

    def _dgrad(self, du, c, res=None, mask=None, in_hw=None):
        """gradient wrt the input of conv `c`; `in_hw` = spatial size of that input (needed for stride 2)."""
        n, ho, wo, _ = du.shape
        mb = getattr(mask, '_cms_relu_bits', None) if mask is not None else None
        if mb is not None and not self.relu_bits:
            mb = None
        if mb is not None:
            mask = None                      # the bits the producing launch wrote instead of the activation itself
        if c.stride == 1:
            return ops.conv_igemm(du, c.wT, c.neg_taps, res=res, mode=1, mask_src=mask, tile=self._tile(c.cin), mask_bits=mb)
        return ops.conv_igemm(du, c.wT, c.neg_taps, res=res, mode=1, mask_src=mask, out_hw=(ho, wo),
                              out_stride=c.stride, out_full_hw=in_hw, tile=self._tile(c.cin), mask_bits=mb)

    def _grad_sentinel(self):
        if getattr(self, '_sentinel', None) is None:
            self._sentinel = dict(self.net.named_parameters())[self.blocks[-1].c3.wkey]
        return self._sentinel

    def _want_w(self):
        data_only = self.data_grad_only or getattr(self.net, '_data_grad_only', False)
        return self.trainable and not data_only and self.arena.grad is not None

    def _refresh_for_backward(self):
        if self._wT_version != self.version or self.blocks[0].c1.wT is None:
            self._refresh_backward_weights()
            self._wT_version = self.version

    def _head_bias_grads(self, dlogits, want_w, side=None):
        """Bias gradients of the two live ASPP branches (plain reduction: torch ops). `side`: the weight-gradient stream they
        are issued on (behind the main stream's work so far) instead of in front of the data-gradient chain."""
        if want_w:
            if side is not None:
                side.wait_stream(torch.cuda.current_stream())
                dlogits.record_stream(side)
            with torch.cuda.stream(side if side is not None else torch.cuda.current_stream()):
                db = dlogits.sum(dim=(0, 2, 3))
                for k in self.aspp_keys:
                    self.arena.view(k + '.bias', self.arena.grad).add_(db)

    def _head_weight_grads(self, dwall, side=None):
        """dWall rows (tap*C + class) -> the (9, C, 2048) gradient tensors of the two branches. `side`: the stream the head's
        weight gradient ran on (the adds must follow it there)."""
        C = self.num_classes
        a = self.arena
        with torch.cuda.stream(side if side is not None else torch.cuda.current_stream()):
            for i, k in enumerate(self.aspp_keys):
                a.packed(k + '.weight', a.grad).add_(dwall[0, 9 * C * i:9 * C * (i + 1)].view(9, C, 2048))

    def _finish_head(self, dwall, side):
        """The head's weight gradient into the arena, on the stream that computed it; a pass that joins its weight-gradient
        streams itself (no deferred join) makes the current stream wait for these adds too."""
        self._head_weight_grads(dwall, side)
        if side is not None and not self.defer_wgrad_join:
            torch.cuda.current_stream().wait_stream(side)

    def bucket_starts(self):
        """Bottleneck indices at which a gradient bucket of the data-parallel all-reduce closes (step.GradBuckets):
        [layer4 + head], the two halves of layer3, [layer1 - layer2]; the stem's slice follows the autograd backward."""
        l2, l3, l4 = self._layer_first[1], self._layer_first[2], self._layer_first[3]
        # (round 5: layer2's first bottleneck closes a slice too -- the weight-gradient streams meet there, which is what an early
        # optimizer launch over [layer2 .. head] needs, step._arm_early_optimizer with CMS_TAIL_OPT_CUT=l2)
        return sorted(set([0, l2, l3, (l3 + l4 + 1) // 2, l4]))

    def _backward_chain(self, saved, dlg, want_w, sides, hook, box=None):
        """The launches of the backward pass (recordable): ASPP head weight + data gradients, then the bottlenecks
        from the last to the first. `dlg`: fp32 (N, C, h, w) logit gradient. `hook(bi)` is called on the weight-gradient
        stream right after the weight gradients of bottleneck `bi` were issued. -> (dx, dwall or None)"""
        x4 = saved[-1]
        # Weight gradients only feed the optimizer, the data-gradient chain never waits for them: they run on a second
        # HIP stream, one bottleneck behind the chain, and fill the tail / memory-wait gaps of the dgrad launches.
        main = torch.cuda.current_stream()
        # head: D[n][y][x][tap*C + c] = dlogits[n][c][y - dy][x - dx]; dX = D . Wall, dWall = D^T . X (csrc/aspp.hip)
        d = ops.aspp_spread_bwd(dlg, self.aspp_taps, self.aspp_zc, self.dtype)
        dwall = None
        keep = []                 # tensors read on the side stream must outlive the python scope that made them
        if want_w:
            dwall = torch.empty((1, self.aspp_zc, 2048), dtype=torch.float32, device=d.device)
            # (round 5) the head's weight gradient (91 us at cfg 2) feeds nothing but the optimizer: on the first weight-gradient
            # stream, like every other weight gradient, instead of in front of the data-gradient chain; CMS_HEAD_WGRAD_SIDE=0: A/B
            hs = sides[0] if (sides and self.head_wgrad_side) else None
            if hs is not None:
                ops.stream_wait(hs, main)
                keep.append((d, dwall))
            with torch.cuda.stream(hs if hs is not None else main):
                ops.memset_zero(dwall)
                ops.conv_wgrad(d, x4, [(0, 0)], dwall)
            if box is not None:
                box['dwall'] = dwall
        x4b = getattr(x4, '_cms_relu_bits', None) if self.relu_bits else None
        dC = ops.conv_igemm(d, self.aspp_wallT, [(0, 0)], mode=1, mask_src=None if x4b is not None else x4, mask_bits=x4b)
        capture = getattr(self, 'debug_capture', None)
        closes = set(self.bucket_starts())
        pending, pending_blocks = [], []
        for bi in range(len(self.blocks) - 1, -1, -1):
            if capture is not None:
                capture[bi] = dC
            b = self.blocks[bi]
            xin, a1, a2 = saved[bi]
            in_hw = (xin.shape[1], xin.shape[2])
            dU2 = self._dgrad(dC, b.c3, mask=a2)
            dU1 = self._dgrad(dU2, b.c2, mask=a1)
            if not want_w:
                pass
            elif sides and self.wgrad_group_blocks > 0 and self.dtype == torch.bfloat16:
                # grouped: collect the launches of this bottleneck; the group goes out behind the data gradients of its LAST
                # bottleneck (a gradient bucket closing here ends the group too)
                keep.append((dC, dU2, dU1))
                pending_blocks.append(bi)
                pending += [(dC, a2, b.c3), (dU2, a1, b.c2)] + ([(dC, xin, b.cd)] if b.cd is not None else []) + [(dU1, xin, b.c1)]
                if len(pending_blocks) >= self.wgrad_group_blocks or bi in closes or bi == 0:
                    ops.stream_wait(sides[0], main)
                    with torch.cuda.stream(sides[0]):
                        grouped = [(du_, x_, c_.taps, self.arena.packed(c_.wkey, self.arena.grad), c_.stride, c_.scale)
                                   for du_, x_, c_ in pending if c_.wdot is None]
                        ops.conv_wgrad_group(grouped)
                        for du_, x_, c_ in pending:
                            if c_.wdot is not None:
                                self._wgrad(du_, x_, c_)
                        for pb in pending_blocks:
                            hook(pb)
                    pending, pending_blocks = [], []
            elif sides:
                for sd in sides:
                    ops.stream_wait(sd, main)
                keep.append((dC, dU2, dU1))
                jobs = [(dC, a2, b.c3), (dU2, a1, b.c2)] + ([(dC, xin, b.cd)] if b.cd is not None else []) + [(dU1, xin, b.c1)]
                for ji, (du_, x_, c_) in enumerate(jobs):
                    with torch.cuda.stream(sides[ji % len(sides)]):
                        self._wgrad(du_, x_, c_)
                if len(sides) > 1 and bi in closes:
                    for sd in sides[1:]:                      # a bucket closes here: stream 0 must have seen every stream
                        ops.stream_wait(sides[0], sd)
                with torch.cuda.stream(sides[0]):
                    hook(bi)
            else:
                self._block_wgrads(b, dC, dU2, dU1, xin, a1, a2)
                hook(bi)
            dres = dC if b.cd is None else self._dgrad(dC, b.cd, in_hw=in_hw)
            dC = self._dgrad(dU1, b.c1, res=dres, mask=None if bi == 0 else xin, in_hw=in_hw)
        if not self.defer_wgrad_join:
            for sd in sides:
                ops.stream_wait(main, sd)
        self._retire_keep(keep)
        return dC, dwall

    def backward(self, token, dlogits):
        """dlogits fp32 (N,C,h,w); `token` from `forward(.., save=True)`. Accumulates weight gradients into the arena;
        returns d loss / d x (NHWC, the executor's dtype)."""
        bn = (token[0].bn if self.use_programs and isinstance(token, tuple) and isinstance(token[0], ops.Program)
              else len(token[0]) == 7)
        chain = self._backward_chain_bn if bn else self._backward_chain
        if bn:
            if self.__dict__.get('_wT_raw_version', -1) != self.version or self.blocks[0].c1.wT_raw is None:
                self._refresh_aspp_fwd()
                self._refresh_backward_weights_bn()
                self._wT_raw_version = self.version
        else:
            self._refresh_for_backward()
        want_w = self._want_w()
        if want_w and self._grad_sentinel().grad is None:        # somebody called module.zero_grad(): re-home the views
            self.arena.ensure_grads_attached()
        n, _, h, w = dlogits.shape
        main = torch.cuda.current_stream()
        sides = self._side_streams(self.wgrad_streams) if (self.overlap_wgrad and want_w) else []
        hside = sides[0] if (sides and self.head_wgrad_side and not bn) else None     # where the head's gradients are formed
        if not self.use_programs:
            if self.defer_wgrad_join and sides:
                self._pending_join = list(sides)
            self._head_bias_grads(dlogits, want_w, hside)
            box = {}

            def hook(bi):
                # first call (last bottleneck, on the weight-gradient stream, which has waited for the head's launches):
                # the head's weight gradients go into the arena BEFORE any bucket of it is all-reduced
                if 'done' not in box and box.get('dwall') is not None:
                    self._head_weight_grads(box['dwall'])
                    box['done'] = True
                if self.grad_hook is not None:
                    self.grad_hook(bi)
            # the chain creates dwall before its first hook call: hand it over through the box
            dx, dwall = chain(token, dlogits, want_w, sides, hook, box)
            if dwall is not None and 'done' not in box:
                self._finish_head(dwall, hside)
            return dx
        fprog, gen = token
        if fprog.generation != gen:
            raise RuntimeError('the activations of this forward pass were overwritten by a later forward pass of the '
                               'same shape through the same executor (programs keep ONE set of buffers per shape): '
                               'run backward before the next forward, or set executor.use_programs = False')
        key = (want_w, len(sides), bool(self.defer_wgrad_join))
        if self.defer_wgrad_join and sides:
            self._pending_join = list(sides)
        prog = fprog.bwd.get(key)
        if prog is None:
            prog = ops.Program()
            dlg = torch.empty(tuple(dlogits.shape), dtype=torch.float32, device=dlogits.device)   # persistent input
            streams = [main] + sides
            with ops.recording(prog, streams):
                dx, dwall = chain(fprog.saved, dlg, want_w, sides, prog.mark)
            prog.dlg, prog.dx, prog.dwall = dlg, dx, dwall
            fprog.bwd[key] = prog
        prog.dlg.copy_(dlogits)
        self._head_bias_grads(dlogits, want_w, hside)
        streams = [main] + sides
        if self.grad_hook is None or not want_w:
            prog.run(streams)
            if prog.dwall is not None:
                self._finish_head(prog.dwall, hside)     # (behind the head's weight gradient, on the stream it ran on)
        else:
            # segments between the recorded block marks: the hook (bucketed all-reduce) is host work that must see the
            # weight-gradient stream as its current stream, right after the weight gradients of its block
            first = 0
            hook_stream = sides[0] if sides else main
            head_done = False
            wanted = getattr(self.grad_hook, 'blocks', None)      # a hook that only acts at some bottlenecks: fewer segments
            for idx, bi in prog.marks:
                if wanted is not None and bi not in wanted:
                    continue
                prog.run(streams, first, idx)
                first = idx
                with torch.cuda.stream(hook_stream):
                    if not head_done and prog.dwall is not None:
                        # the head's weight gradients reach the arena BEFORE any bucket of it is all-reduced (the
                        # weight-gradient stream has waited for the head's launches at this point)
                        self._head_weight_grads(prog.dwall)
                        head_done = True
                    self.grad_hook(bi)
            prog.run(streams, first, -1)
            if prog.dwall is not None and not head_done:
                self._finish_head(prog.dwall, hside)
        self._account(prog)
        return prog.dx.clone()

    def _retire_keep(self, keep):
        """End of a backward chain: the gradient tensors the weight-gradient streams read. Joined chain (or a recording: the
        program owns the buffers): they may go now. EAGER chain with a deferred join: they are main-stream
        allocations still being read on the side streams -- freed here the caching allocator could hand their blocks to the
        max-pool / stem backward that follows on the main stream; they stay alive until `join_wgrad()` has made the main
        stream wait for the side streams."""
        if keep and self.defer_wgrad_join and ops._REC is None:
            self._pending_keep = (self.__dict__.get('_pending_keep') or []) + [keep]

    def join_wgrad(self):
        """The current stream waits for the weight-gradient stream(s) of the last backward pass (see `defer_wgrad_join`)."""
        sides, self._pending_join = self._pending_join, None
        if sides:
            main = torch.cuda.current_stream()
            for sd in sides:
                main.wait_stream(sd)
        self._pending_keep = None          # (after the waits: frees are ordered behind the side streams' reads)

    def _block_wgrads(self, b, dC, dU2, dU1, xin, a1, a2):
        self._wgrad(dC, a2, b.c3)
        self._wgrad(dU2, a1, b.c2)
        if b.cd is not None:
            self._wgrad(dC, xin, b.cd)
        self._wgrad(dU1, xin, b.c1)

    def _side_streams(self, k):
        k = max(1, min(int(k), 3))
        while len(self._sides) < k:
            self._sides.append(ops.pooled_stream(self.arena.device, 'wgrad{}'.format(len(self._sides))))
        return self._sides[:k]

    def _fwd_side_stream(self):
        if self.__dict__.get('_fwd_side') is None:
            self._fwd_side = ops.pooled_stream(self.arena.device, 'fwd_half_{}'.format('s' if self.trainable else 't'))
        return self._fwd_side

    def _side_stream(self):
        if self._side is None:
            self._side = ops.pooled_stream(self.arena.device, 'side')
        return self._side


class DeepLabV3PlusBackboneExecutor(DeepLabHipExecutor):
    """
    Executor of the DeepLab v3+ backbone (torchvision-style ResNet-101 v1.5, layer3 / layer4 dilated, frozen BatchNorm
    statistics) on the MFMA convolution kernels, forward and backward. Returns the two taps the head consumes
    (architectures/deeplab3plus.py:96-98). What differs from the DeepLab v2 body:

      * the 3x3 carries the stride (`layer2.0.conv2`): its data gradient is a true transposed convolution, the one
        launch of the chain that goes through the library;
      * the BatchNorm affine parameters TRAIN although the statistics are frozen (torchvision leaves them trainable,
        `freeze_batchnorm()` only switches the statistics): with G = sum_p dU x the unscaled weight gradient,
        d(bias) = sum_p dU and d(weight) = (<W, G> - mean * d(bias)) / sqrt(var + eps) -- both come out of the weight
        gradient kernel as side outputs (cms_wgrad_desc.wdot / dbeta), exact and without dividing by the weight;
      * two outputs (layer1 -> 'low_level', layer4 -> 'out'), so the backward chain takes a second gradient in at the
        layer1 / layer2 boundary.
    """

    def __init__(self, wrapper, dtype=torch.bfloat16):
        self._init_common(wrapper, dtype)
        # passes recorded once per input shape and replayed (csrc/program.hip). The fp32 PARITY configuration with a
        # trainable BatchNorm affine derives d(gamma) / d(beta) with tensor ops between the launches (the f32 weight-
        # gradient kernel has no side outputs): issued launch by launch
        self.use_programs = not (dtype == torch.float32 and self.bn_trainable and self.trainable)
        # torchvision's stem: same 7x7/2 convolution, max-pool WITHOUT ceil_mode, trainable BatchNorm affine
        self.stem_wkey, self.stem_bn, self.stem_ceil = 'deeplab.backbone.conv1.weight', 'deeplab.backbone.bn1', False
        self.stem_w147 = None
        self._stem_version = -1
        bb = wrapper.deeplab.backbone
        self._add_blocks('deeplab.backbone.', [bb['layer{}'.format(li)] for li in range(1, 5)])
        self.tap_low = self._layer_first[1] - 1          # last bottleneck of layer1
        self._phase_w = {}                               # (conv id, py, px) -> persistent sub-weight of the phase

    def _prepare_forward(self):
        if not self._affine_ready:
            self._refresh_affine()

    def _taps_pass(self, x, save):
        st = self.fwd_begin(x, save)
        low = None
        for bi in range(len(self.blocks)):
            if ops._REC is not None:
                ops._REC[0].group = bi
            self.fwd_block(st, bi)
            if bi == self.tap_low:
                low = st['cur']
        if save:
            st['saved'].append(st['cur'])
        return low, st['cur'], st['saved']

    def taps_program(self, shape, save):
        """The recorded backbone pass for a stem output of `shape`: x_in (persistent input), low, out, saved."""
        key = ('taps', tuple(int(v) for v in shape), bool(save), self._tile_key())
        prog = self._program_lookup(key)
        if prog is None:
            self._prepare_forward()
            prog = ops.Program()
            x_in = torch.empty(tuple(shape), dtype=self.dtype, device=self.arena.device)
            with ops.recording(prog, [torch.cuda.current_stream()]):
                low, out, saved = self._taps_pass(x_in, save)
            prog.x_in, prog.low, prog.out, prog.saved = x_in, low, out, saved
            prog.bwd = {}
            prog.generation = -1
            self._programs[key] = prog
        return prog

    def forward_taps(self, x, save=False):
        """x: bf16 NHWC stem output -> (low_level (N,h/4,w/4,256), out (N,h/8,w/8,2048)) bf16 NHWC [, saved].
        With programs the two taps are the program's own buffers: valid until the next pass of the same shape through
        this executor is ENQUEUED behind their consumers (stream order), which is how the head uses them."""
        if not self.use_programs or self._eager_this_time('taps', x.shape, save):
            low, out, saved = self._taps_pass(x, save)
            return (low, out, saved) if save else (low, out)
        prog = self.taps_program(x.shape, save)
        self._prepare_forward()
        prog.x_in.copy_(x)
        prog.run([torch.cuda.current_stream()])
        self._stamp(prog)
        # (fresh tensor objects over the program's buffers: an autograd node must not hand out the same object twice)
        if save:
            return prog.low.detach(), prog.out.detach(), (prog, prog.generation)
        return prog.low.detach(), prog.out.detach()

    def _refresh_backward_weights(self):
        super()._refresh_backward_weights()
        for (cid, py, px), (c, idx, buf) in self._phase_w.items():          # persistent phase sub-weights, in place
            torch.index_select(c.wT, 0, idx, out=buf)

    def _dgrad_strided(self, du, c, mask, in_hw):
        """Data gradient of the stride-2 3x3 convolution (torchvision v1.5 `layer2.0.conv2`; + ReLU mask of its input):
        a transposed convolution, computed as its four PHASES on the MFMA kernel. With y[o] = sum_k W[k] x[2o + k - 1],
        an input pixel 2a + p (p = its parity) receives from the taps k = p + 1 (mod 2): p = 0 -> k = 1 (dy[a]);
        p = 1 -> k = 0 (dy[a + 1]) and k = 2 (dy[a]). Each (py, px) phase is therefore a stride-1 convolution over the dy
        grid with 1, 2, 2 or 4 taps whose outputs land on every second pixel of dx starting at (py, px). The phases'
        sub-weights live in persistent buffers (refreshed with the dgrad operands), so that a recorded pass stays valid."""
        if not (c.stride == 2 and c.ksize == 3 and c.pad == 1 and c.dil == 1):
            raise NotImplementedError('phase decomposition is written for the 3x3 / stride 2 / pad 1 convolution')
        n = du.shape[0]
        H, W = int(in_hw[0]), int(in_hw[1])
        dx = torch.empty((n, H, W, c.cin), dtype=du.dtype, device=du.device)
        sel = {0: [1], 1: [0, 2]}
        for py in (0, 1):
            for px in (0, 1):
                ha, wb = (H - py + 1) // 2, (W - px + 1) // 2
                if ha <= 0 or wb <= 0:
                    continue
                ks = [(ky, kx) for ky in sel[py] for kx in sel[px]]
                ent = self._phase_w.get((id(c), py, px))
                if ent is None:
                    idx = _index_tensor(tuple(ky * 3 + kx for ky, kx in ks), du.device)
                    ent = (c, idx, c.wT[idx].contiguous())                   # (taps, Cin, Cout), BN scale folded
                    self._phase_w[(id(c), py, px)] = ent
                taps = [((py + 1 - ky) // 2, (px + 1 - kx) // 2) for ky, kx in ks]
                ops.conv_igemm(du, ent[2], taps, mode=1, mask_src=mask, out=dx, out_hw=(ha, wb), out_stride=2,
                               out_full_hw=(H, W), out_pixel_offset=py * W + px)
        return dx

    def _taps_chain(self, saved, dC, d_low, track_bn, side, rec):
        """The launches of the backward pass (recordable). `dC`: gradient wrt the layer4 output, already masked with its
        ReLU. `d_low`: gradient wrt the layer1 tap or None; while recording it is added between two program segments
        (`rec.dres` is the buffer it goes into, the segment boundary is marked 'dlow')."""
        main = torch.cuda.current_stream()
        if track_bn:
            ops.memset_zero(self._wdot_all)
            ops.memset_zero(self._dbeta_all)
        keep = []
        wjob = 0
        capture = getattr(self, 'debug_capture', None)      # {block: (dC, dU2, dU1)} of the pass being issued (diagnostics)
        for bi in range(len(self.blocks) - 1, -1, -1):
            if rec is not None:
                rec.group = bi
            b = self.blocks[bi]
            xin, a1, a2 = saved[bi]
            in_hw = (xin.shape[1], xin.shape[2])
            dU2 = self._dgrad(dC, b.c3, mask=a2)
            if b.c2.stride == 1:
                dU1 = self._dgrad(dU2, b.c2, mask=a1)
            else:
                dU1 = self._dgrad_strided(dU2, b.c2, a1, (a1.shape[1], a1.shape[2]))
            if capture is not None:
                capture[bi] = (dC, dU2, dU1)
            if isinstance(side, (list, tuple)) and len(side) > 1 and os.environ.get('CMS_V3_WGRAD_PER_JOB', '1') != '0':
                # (round 6) two weight-gradient streams, the JOBS alternating over them with a counter that runs across the
                # blocks: per-block alternation left one stream 2.4 ms behind the other at the end of the pass (layer 1's three
                # blocks at 129 x 129: two of them on one stream) with the main stream idle behind both (profiles/r06al_*)
                for sd in side:
                    ops.stream_wait(sd, main)
                keep.append((dC, dU2, dU1))
                jobs = [(dC, a2, b.c3), (dU2, a1, b.c2)] + ([(dC, xin, b.cd)] if b.cd is not None else []) + [(dU1, xin, b.c1)]
                for du_, x_, c_ in jobs:
                    with torch.cuda.stream(side[wjob % len(side)]):
                        self._wgrad(du_, x_, c_)
                    wjob += 1
            elif side is not None:
                # `side`: one stream, or a list the blocks alternate over (CMS_V3_WGRAD_PER_JOB=0)
                sd = side[bi % len(side)] if isinstance(side, (list, tuple)) else side
                ops.stream_wait(sd, main)
                keep.append((dC, dU2, dU1))
                with torch.cuda.stream(sd):
                    self._block_wgrads(b, dC, dU2, dU1, xin, a1, a2)
            else:
                self._block_wgrads(b, dC, dU2, dU1, xin, a1, a2)
            dres = dC if b.cd is None else self._dgrad(dC, b.cd, in_hw=in_hw)
            if bi == self.tap_low + 1 and d_low is not None:
                # second gradient into the layer1 output (masked with it just below). layer2.0 has a downsample
                # convolution, so dres is that convolution's own output buffer: the in-place add races with nobody
                if b.cd is None:
                    raise RuntimeError('the layer1 tap must feed a bottleneck with a downsample branch')
                if rec is not None:
                    rec.mark('dlow')
                    rec.dres = dres
                else:
                    dres.add_(d_low)
            dC = self._dgrad(dU1, b.c1, res=dres, mask=None if bi == 0 else xin, in_hw=in_hw)
        if side is not None:
            sds = list(side) if isinstance(side, (list, tuple)) else [side]
            for sd in sds[1:]:
                ops.stream_wait(sds[0], sd)
            with torch.cuda.stream(sds[0]):
                self._wfinish_flush()
            for sd in sds:
                ops.stream_wait(main, sd)
        else:
            self._wfinish_flush()
        del keep
        return dC

    def backward_taps(self, saved, d_low, d_out):
        """Gradients wrt the two taps (bf16 NHWC or None) -> gradient wrt the stem output; weight and BatchNorm-affine
        gradients are accumulated into the arena."""
        if self._wT_version != self.version or self.blocks[0].c1.wT is None:
            self._refresh_backward_weights()
            self._wT_version = self.version
        track_bn = self.bn_trainable
        main = torch.cuda.current_stream()
        side = self._side_stream() if self.overlap_wgrad else None
        # (round 6) TWO weight-gradient streams, the blocks alternating over them: with the wide layers on the eight-phase kernel
        # (`_wfinish_takes`: ~56 workgroups per launch) one stream leaves half the machine idle beside the data-gradient chain --
        # 148 img/s on one stream, 174 on two, 157.5 with the side-output kernel on one (cfg 4, profiles/r06aj_*). With the side-output
        # kernel everywhere (CMS_V3_WGRAD8=0) two streams lose (152.8 vs 154.7, profiles/r06o_*): one stream then. CMS_V3_WGRAD_STREAMS overrides.
        two = self.dtype == torch.bfloat16 and not ops.deterministic_wgrad() and os.environ.get('CMS_V3_WGRAD8', '1') != '0'
        nws = os.environ.get('CMS_V3_WGRAD_STREAMS', '2' if two else '1')
        if side is not None and nws in ('2', '3'):
            side = [side] + [ops.pooled_stream(self.arena.device, 'wgrad{}'.format(i)) for i in range(1, int(nws))]
        recorded = isinstance(saved, tuple) and len(saved) == 2 and isinstance(saved[0], ops.Program)
        if not recorded:
            x4 = saved[-1]
            if d_out is None:
                d_out = torch.zeros_like(x4)
            dC = (d_out * (x4 > 0)).contiguous()
            dx = self._taps_chain(saved, dC, d_low, track_bn, side, None)
        else:
            fprog, gen = saved
            if fprog.generation != gen:
                raise RuntimeError('the activations of this forward pass were overwritten by a later forward pass of the '
                                   'same shape through the same executor (programs keep ONE set of buffers per shape): '
                                   'run backward before the next forward, or set executor.use_programs = False')
            x4 = fprog.saved[-1]
            key = (d_low is not None, track_bn, side is not None, isinstance(side, list))
            prog = fprog.bwd.get(key)
            streams = [main] + (list(side) if isinstance(side, list) else ([side] if side is not None else []))
            if prog is None:
                prog = ops.Program()
                prog.dC_in = torch.empty_like(x4)
                prog.dlow_in = torch.empty_like(fprog.low) if d_low is not None else None
                prog.dres = None
                with ops.recording(prog, streams):
                    prog.dx = self._taps_chain(fprog.saved, prog.dC_in, prog.dlow_in, track_bn, side, prog)
                fprog.bwd[key] = prog
            if d_out is None:
                prog.dC_in.zero_()
            else:
                torch.mul(d_out, x4 > 0, out=prog.dC_in)
            if d_low is not None:
                prog.dlow_in.copy_(d_low)
                cut = [i for i, tag in prog.marks if tag == 'dlow'][0]
                prog.run(streams, 0, cut)
                prog.dres.add_(prog.dlow_in)              # main stream, between the two segments
                prog.run(streams, cut, -1)
            else:
                prog.run(streams)
            self._account(prog)
            dx = prog.dx.clone()
        if track_bn:
            a, ix = self.arena, self._bn_idx
            dgamma = (self._wdot_all - a.flat[ix['running_mean']] * self._dbeta_all) * \
                torch.rsqrt(a.flat[ix['running_var']] + 1e-5)
            a.grad.index_add_(0, ix['weight'], dgamma)
            a.grad.index_add_(0, ix['bias'], self._dbeta_all)
        return dx

    def backward(self, saved, dlogits):
        raise NotImplementedError('use backward_taps')


class _V3BodyFn(torch.autograd.Function):
    """DeepLab v3+ backbone (after the stem) as one autograd node with two outputs."""

    @staticmethod
    def forward(ctx, x_nhwc, executor, need_grad):
        if need_grad:
            low, out, saved = executor.forward_taps(x_nhwc, save=True)
        else:
            low, out = executor.forward_taps(x_nhwc, save=False)
            saved = None
        ctx.executor = executor
        ctx.saved_acts = saved
        return low, out

    @staticmethod
    def backward(ctx, d_low, d_out):
        dt = ctx.executor.dtype
        dx = ctx.executor.backward_taps(ctx.saved_acts,
                                        None if d_low is None else d_low.contiguous().to(dt),
                                        None if d_out is None else d_out.contiguous().to(dt))
        ctx.saved_acts = None
        return dx, None, None


def run_v3_body(executor, x_nhwc):
    need_grad = torch.is_grad_enabled() and executor.trainable
    if need_grad and not x_nhwc.requires_grad:
        x_nhwc = x_nhwc.detach().requires_grad_(True)
    return _V3BodyFn.apply(x_nhwc, executor, need_grad)


def _wbuf_of(arena, dtype):
    """Flat operand buffer of `dtype`: the bf16 copy the fused optimizer maintains, or the fp32 master arena itself."""
    return arena.bf16 if dtype == torch.bfloat16 else arena.flat


_MAX_TAPS = 18            # CMS_CONV_MAX_TAPS: larger kernels (7 x 7) are issued as chunks of taps that accumulate


def _tap_chunks(n):
    return [(t0, min(n, t0 + _MAX_TAPS)) for t0 in range(0, n, _MAX_TAPS)]


def _pad64(c):
    return (c + 63) // 64 * 64


class _HipConvGeneralFn(torch.autograd.Function):
    """`conv(x)` for ANY bias-free, ungrouped nn.Conv2d with a square kernel and symmetric stride / padding / dilation on
    the hand-written MFMA kernels (csrc/conv.hip for bf16, csrc/conv_f32.hip for the fp32 parity configuration), forward,
    data gradient and weight gradient -- the engine of the networks that run layer by layer (DeepLab v3+ head, the
    U-Nets' encoders and decoders: architectures/deeplab3plus.py:26-101, resunet.py:36-108, denseunet.py:36-143).

      * channel counts are zero-padded to multiples of 64 for the call (DenseNet's 48-multiples, the 304-channel concat
        of the v3+ decoder, 3-channel images); the weight gradient of a padded layer comes back through an fp32 scratch;
      * kernels with more than 18 taps (the 7 x 7 stems) run as chunks of taps, each launch adding to the previous one
        through the residual input of the epilogue;
      * strided convolutions gather with `stride` in the forward pass; their data gradient is the transposed convolution
        as stride x stride PHASES (pixel s*a + p receives the taps k with (p + pad - k*dil) % s == 0 from dy[a + (p + pad -
        k*dil) / s]), each a stride-1 launch scattered to every s-th pixel.
    """

    @staticmethod
    def forward(ctx, x, weight, arena, key, geom, dtype):
        k, stride, pad, dil = geom
        n, cin, h, w = (int(v) for v in x.shape)
        wp = arena.packed(key, _wbuf_of(arena, dtype))                 # (taps, Cout, Cin)
        cout = int(wp.shape[1])
        cpad, opad = _pad64(cin), _pad64(cout)
        if cpad != cin:
            xh = torch.zeros((n, h, w, cpad), dtype=dtype, device=x.device)
            xh[..., :cin] = x.permute(0, 2, 3, 1)
        else:
            xh = x.permute(0, 2, 3, 1).contiguous().to(dtype)
        if cpad != cin or opad != cout:
            # (round 6) the zero-padded operand is made once per weight VERSION, not once per call (two launches per convolution
            # of DenseNet-161's 48-multiples: the VAT iteration is launch-bound)
            def _pad():
                t = torch.zeros((wp.shape[0], opad, cpad), dtype=dtype, device=x.device)
                t[:, :cout, :cin] = wp
                return t
            wpad = arena.cached(key, 'pad', dtype, _pad)
        else:
            wpad = wp
        taps = ops.conv_taps(k, k, dil, pad)
        ho = (h + 2 * pad - dil * (k - 1) - 1) // stride + 1
        wo = (w + 2 * pad - dil * (k - 1) - 1) // stride + 1
        y = None
        for t0, t1 in _tap_chunks(len(taps)):
            wc = wpad if (t0 == 0 and t1 == len(taps)) else wpad[t0:t1].contiguous()
            y = ops.conv_igemm(xh, wc, taps[t0:t1], stride=stride, out_hw=(ho, wo), res=y)
        ctx.arena, ctx.key, ctx.geom, ctx.dtype = arena, key, geom, dtype
        ctx.wversion = arena.version
        ctx.cin, ctx.cout, ctx.in_hw = cin, cout, (h, w)
        ctx.need_w = weight.requires_grad and arena.grad is not None
        ctx.save_for_backward(xh, wpad)
        if opad != cout:
            y = y[..., :cout].contiguous()
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        xh, wpad = ctx.saved_tensors
        a, cin, cout, dtype = ctx.arena, ctx.cin, ctx.cout, ctx.dtype
        k, stride, pad, dil = ctx.geom
        h, w = ctx.in_hw
        ntaps, opad, cpad = (int(v) for v in wpad.shape)
        n, ho, wo = int(dy.shape[0]), int(dy.shape[2]), int(dy.shape[3])
        if opad != cout:
            dyh = torch.zeros((n, ho, wo, opad), dtype=dtype, device=dy.device)
            dyh[..., :cout] = dy.permute(0, 2, 3, 1)
        else:
            dyh = dy.permute(0, 2, 3, 1).contiguous().to(dtype)
        taps = ops.conv_taps(k, k, dil, pad)
        if ctx.need_w:
            # (round 6) the weight gradient FIRST and on a side stream (ops.layer_wgrad_stream): it feeds nothing but the optimizer
            # and runs beside the data gradient below instead of behind it
            padded = opad != cout or cpad != cin
            # (only where the launch is worth a stream switch: >= 2 GFLOP; the small layers of the U-Nets are host-bound and the
            # fork / join / record_stream calls cost them more than the overlap gives -- DenseNet-161 VAT: 46.0 vs 47.6 img/s)
            big = 2.0 * n * ho * wo * opad * cpad * ntaps >= 2e9
            side = ops.layer_wgrad_stream(dyh.device, (id(a), ctx.key)) if big else None

            def _wgrad():
                dw = torch.zeros(wpad.shape, dtype=torch.float32, device=dyh.device) if padded else a.packed(ctx.key, a.grad)
                for t0, t1 in _tap_chunks(ntaps):
                    ops.conv_wgrad(dyh, xh, taps[t0:t1], dw[t0:t1], stride=stride, wg_target=0 if side is None else 128)
                if padded:
                    a.packed(ctx.key, a.grad).add_(dw[:, :cout, :cin])
            if side is None:
                _wgrad()
            else:
                with torch.cuda.stream(side):
                    _wgrad()
                dyh.record_stream(side)
                xh.record_stream(side)
        dx = None
        if ctx.needs_input_grad[0]:
            mk_T = lambda: ops.conv_pack_transpose(wpad, flip=False, out_dtype=torch.float32 if dtype == torch.float32 else None)
            # the transposed operand of the weights this pass's FORWARD used: cached while they are still the current version
            # (the VAT direction pass and the teacher's passes run several backward / forward passes per optimizer step)
            wT = a.cached(ctx.key, 'padT', dtype, mk_T) if a.version == ctx.wversion else mk_T()
            if stride == 1:
                dxp = None
                for t0, t1 in _tap_chunks(ntaps):
                    wc = wT if (t0 == 0 and t1 == ntaps) else wT[t0:t1].contiguous()
                    dxp = ops.conv_igemm(dyh, wc, [(-ty, -tx) for ty, tx in taps[t0:t1]], mode=1, res=dxp)
            else:
                dxp = torch.zeros((n, h, w, cpad), dtype=dtype, device=dy.device)
                for py in range(stride):
                    for px in range(stride):
                        ha, wb = (h - py + stride - 1) // stride, (w - px + stride - 1) // stride
                        if ha <= 0 or wb <= 0:
                            continue
                        sel = [(ky, kx) for ky in range(k) for kx in range(k)
                               if (py + pad - ky * dil) % stride == 0 and (px + pad - kx * dil) % stride == 0]
                        if not sel:
                            continue                       # pixels of this phase receive nothing: they stay zero
                        idx = _index_tensor(tuple(ky * k + kx for ky, kx in sel), dy.device)
                        wsub = wT.index_select(0, idx)
                        offs = [((py + pad - ky * dil) // stride, (px + pad - kx * dil) // stride) for ky, kx in sel]
                        first = True
                        for t0, t1 in _tap_chunks(len(sel)):
                            wc = wsub if (t0 == 0 and t1 == len(sel)) else wsub[t0:t1].contiguous()
                            ops.conv_igemm(dyh, wc, offs[t0:t1], mode=1, out=dxp, out_hw=(ha, wb), out_stride=stride,
                                           out_full_hw=(h, w), out_pixel_offset=py * w + px, res=None if first else dxp)
                            first = False
            dx = (dxp[..., :cin] if cpad != cin else dxp).permute(0, 3, 1, 2)
        return dx, None, None, None, None, None


_INDEX_TENSORS = {}


def _index_tensor(values, device):
    """A small int64 index tensor on the device, made ONCE per (values, device): a host -> device copy from pageable memory
    synchronises, which a pass being captured into a hipGraph may not do (vat.VATMeanTeacherStep._graphed_grads)."""
    key = (values, str(device))
    t = _INDEX_TENSORS.get(key)
    if t is None:
        t = _INDEX_TENSORS[key] = torch.tensor(list(values), dtype=torch.long, device=device)
    return t


class _HipClassifierFn(torch.autograd.Function):
    """A 1x1 classifier with bias (DeepLab v3+ head, deeplab3plus.py:47: 256 -> num_classes; the U-Nets' final_clf): class
    axis padded to 64 for the MFMA kernels, fp32 NCHW logits straight out of the convolution epilogue."""

    @staticmethod
    def forward(ctx, x, weight, bias, arena, wkey, bkey, dtype):
        n, cin, h, w = x.shape
        xh = x.permute(0, 2, 3, 1).contiguous().to(dtype)
        wp = arena.packed(wkey, _wbuf_of(arena, dtype))                # (1, C, Cin)
        c = int(wp.shape[1])
        wpad = torch.zeros((1, 64, cin), dtype=dtype, device=x.device)
        wpad[:, :c] = wp
        bpad = torch.zeros(64, dtype=torch.float32, device=x.device)
        bpad[:c] = arena.view(bkey)
        logits = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
        ops.conv_igemm(xh, wpad, [(0, 0)], bias=bpad, out_f32_nchw=logits, cout_real=c)
        ctx.arena, ctx.wkey, ctx.bkey, ctx.c, ctx.dtype = arena, wkey, bkey, c, dtype
        ctx.need_w = weight.requires_grad and arena.grad is not None
        ctx.save_for_backward(xh, wpad)
        return logits

    @staticmethod
    def backward(ctx, dl):
        xh, wpad = ctx.saved_tensors
        a, c, dtype = ctx.arena, ctx.c, ctx.dtype
        dlh = torch.zeros(tuple(xh.shape[:3]) + (64,), dtype=dtype, device=dl.device)
        dlh[..., :c] = dl.permute(0, 2, 3, 1)
        dx = None
        if ctx.needs_input_grad[0]:
            wT = ops.conv_pack_transpose(wpad, flip=False, out_dtype=torch.float32 if dtype == torch.float32 else None)
            dx = ops.conv_igemm(dlh, wT, [(0, 0)], mode=1).permute(0, 3, 1, 2)
        if ctx.need_w:
            # (round 6) beside the next layers' data gradients, see _HipConvGeneralFn (large launches only)
            big = 2.0 * float(dlh.numel()) * int(xh.shape[3]) >= 2e9
            side = ops.layer_wgrad_stream(dl.device, (id(a), ctx.wkey)) if big else None

            def _wgrad():
                tmp = torch.zeros(wpad.shape, dtype=torch.float32, device=dl.device)
                ops.conv_wgrad(dlh, xh, [(0, 0)], tmp)
                a.packed(ctx.wkey, a.grad).add_(tmp[:, :c])
                a.view(ctx.bkey, a.grad).add_(dl.float().sum(dim=(0, 2, 3)))
            if side is None:
                _wgrad()
            else:
                with torch.cuda.stream(side):
                    _wgrad()
                for t_ in (dlh, xh, dl):
                    t_.record_stream(side)
        return dx, None, None, None, None, None, None


def hip_classifier(x, conv, arena, wkey, bkey, dtype=torch.bfloat16):
    """`conv(x)` for a 1x1 nn.Conv2d WITH bias and <= 64 output channels -> fp32 NCHW."""
    return _HipClassifierFn.apply(x, conv.weight, conv.bias, arena, wkey, bkey, dtype)


def hip_conv_geometry(conv):
    """(k, stride, pad, dil) of an nn.Conv2d the general path can run, or None."""
    kh, kw = conv.kernel_size
    if conv.groups != 1 or kh != kw or conv.stride[0] != conv.stride[1] or conv.padding[0] != conv.padding[1] \
            or conv.dilation[0] != conv.dilation[1] or isinstance(conv.padding, str) \
            or getattr(conv, 'padding_mode', 'zeros') != 'zeros':
        return None
    return int(kh), int(conv.stride[0]), int(conv.padding[0]), int(conv.dilation[0])


def hip_conv2d(x, conv, arena, key, dtype=torch.bfloat16):
    """`conv(x)` WITHOUT its bias (the engines add biases themselves) on the hand-written kernels."""
    geom = hip_conv_geometry(conv)
    if geom is None:
        raise NotImplementedError('no hand-written kernel for this convolution geometry: {}'.format(conv))
    return _HipConvGeneralFn.apply(x, conv.weight, arena, key, geom, dtype)


def _fused_bn_stats():
    """CMS_BN_FUSED_STATS (default 1; A/B switch, read per recording): batch-statistics units take their statistics from the tile
    sums the convolution's epilogue writes (`_fwd_unit_bn`); 0 = the round-3/4 pass over the convolution output."""
    return os.environ.get('CMS_BN_FUSED_STATS', '1') != '0'


def _stem_direct():
    """CMS_STEM_DIRECT (default 1; A/B switch): the stem's max-pool writes into the recorded body pass's input buffer."""
    return os.environ.get('CMS_STEM_DIRECT', '1') != '0'


def _bn_mask_bits():
    """CMS_BN_MASK_BITS (default 1; A/B switch, read per recording): the normalising launch of a batch-statistics unit writes its
    ReLU mask as bits and the unit's backward passes read those instead of y."""
    return os.environ.get('CMS_BN_MASK_BITS', '1') != '0'


def _bn_gate_shortcut():
    """CMS_BN_GATE_SHORTCUT (default 1; A/B switch): batch-statistics backward without the `dres` tensor (`_backward_chain_bn`)."""
    return os.environ.get('CMS_BN_GATE_SHORTCUT', '1') != '0'


def hip_conv2d_eligible(x, conv, dtype=torch.bfloat16):
    """True when the hand-written general path (`hip_conv2d`) can express this convolution: any ungrouped square-kernel convolution
    with symmetric stride / padding / dilation (strided layers as phases, 7 x 7 stems as tap chunks, narrow / odd channel counts
    padded). A library dispatch is not an implementation: since round 6 a layer that fails this test RAISES in every engine (none of
    the reference's networks has one); rounds 2-4 sent stems, strided and narrow layers to MIOpen here."""
    return x.is_cuda and x.dtype == dtype and hip_conv_geometry(conv) is not None


class _StemFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, executor, save=None):
        executor._stem_prepare()
        s = ops.stem_forward(x, executor.stem_w147, executor.stem_scale, executor.stem_bias, executor.dtype)
        # (round 6) the pooled map goes STRAIGHT into the input buffer of the recorded body pass that will consume it, when that pass
        # exists already (every step but the first of a shape): no 17 MB copy per network and step behind the stem
        dst = None
        if save is not None:
            hp, wp = ops._pool_out(int(s.shape[1]), executor.stem_ceil), ops._pool_out(int(s.shape[2]), executor.stem_ceil)
            dst = executor._stem_destination((int(s.shape[0]), hp, wp, int(s.shape[3])), save)
        p, idx = ops.maxpool3x3s2_forward(s, ceil_mode=executor.stem_ceil, out=dst)
        if dst is not None:
            p = dst.detach()                     # a fresh tensor object over the program's buffer (autograd hangs its node on it)
        ctx.executor = executor
        ctx.x_shape = tuple(x.shape)
        ctx.x_dtype = x.dtype
        ctx.w_grad = weight.requires_grad
        ctx.save_for_backward(x, s, idx)
        return p

    @staticmethod
    def backward(ctx, dp):
        x, s, idx = ctx.saved_tensors
        ex = ctx.executor
        ds = ops.maxpool3x3s2_relu_backward(dp.to(s.dtype), idx, s, ceil_mode=ex.stem_ceil)
        if ctx.w_grad and ex._want_w():
            if ex.bn_trainable:
                # frozen statistics, TRAINABLE affine (torchvision-style stem): with G the unscaled weight gradient,
                # d(beta) = sum_p dS and d(gamma) = (<W, G> - mean * d(beta)) / sqrt(var + eps), like the body's convolutions
                a = ex.arena
                g = torch.zeros((49, 64, 3), dtype=torch.float32, device=ds.device)
                ops.stem_wgrad(x, ds, g, None)
                a.packed(ex.stem_wkey, a.grad).add_(g * ex.stem_scale.view(1, -1, 1))
                wdot = (g * a.packed(ex.stem_wkey, ex._wbuf()).float()).sum(dim=(0, 2))
                dbeta = ds.float().sum(dim=(0, 1, 2))
                mean, var = a.view(ex.stem_bn + '.running_mean'), a.view(ex.stem_bn + '.running_var')
                a.view(ex.stem_bn + '.weight', a.grad).add_((wdot - mean * dbeta) * torch.rsqrt(var + 1e-5))
                a.view(ex.stem_bn + '.bias', a.grad).add_(dbeta)
            else:
                ops.stem_wgrad(x, ds, ex.arena.packed(ex.stem_wkey, ex.arena.grad), ex.stem_scale)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.stem_dgrad(ds, ex.stem_w147, ex.stem_scale, ctx.x_shape).to(ctx.x_dtype)
        return dx, None, None, None


class _BodyFn(torch.autograd.Function):
    """The whole body as one autograd node: forward keeps the activations, backward runs the hand-written chain."""

    @staticmethod
    def forward(ctx, x_nhwc, executor, need_grad):
        logits, token = executor.forward(x_nhwc, save=need_grad)
        ctx.executor = executor
        ctx.saved_acts = token
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        dx = ctx.executor.backward(ctx.saved_acts, dlogits.contiguous().float())
        ctx.saved_acts = None
        return dx, None, None


class _BodyPairFn(torch.autograd.Function):
    """Student and teacher bodies in ONE node, their launches interleaved bottleneck by bottleneck: student on the
    current stream, teacher on `side`. Issuing one whole pass after the other leaves the two streams overlapping only
    where the host happens to run ahead of the GPU; interleaved issue keeps two kernels in flight for the whole
    forward pass, which is what fills the launch tails of these small grids."""

    @staticmethod
    def forward(ctx, x_stu, x_tea, ex_stu, ex_tea, side, need_grad):
        if ex_stu.use_programs and ex_tea.use_programs:
            # both passes are recorded programs: ONE native call issues them interleaved, bottleneck by bottleneck
            main = torch.cuda.current_stream()
            ps = ex_stu.forward_program(x_stu.shape, need_grad)
            ex_stu._prepare_forward()
            if x_stu.data_ptr() != ps.x_in.data_ptr():           # (the stem wrote there directly: `_StemFn`)
                ps.x_in.copy_(x_stu)
            with torch.cuda.stream(side):
                pt = ex_tea.forward_program(x_tea.shape, False)
                ex_tea._prepare_forward()
                if x_tea.data_ptr() != pt.x_in.data_ptr():
                    pt.x_in.copy_(x_tea)
            if ps.host_ops or pt.host_ops:
                # SyncBN all-reduces between the launches: the native interleave cannot stop for them -- one pass after the
                # other (each on its stream; they still overlap where the host runs ahead)
                with torch.cuda.stream(side):
                    pt.run([side] + list(getattr(pt, 'extra_streams', [])))
                ps.run([main] + list(getattr(ps, 'extra_streams', [])))
            else:
                ops.run_pair(ps, [main] + list(getattr(ps, 'extra_streams', [])), pt, [side] + list(getattr(pt, 'extra_streams', [])))
            ex_stu._stamp(ps)
            ex_tea._stamp(pt)
            logits_s = ps.logits.clone()
            with torch.cuda.stream(side):
                logits_t = pt.logits.clone()
            saved = (ps, ps.generation) if need_grad else None
        else:
            st_s = ex_stu.fwd_begin(x_stu, need_grad)
            with torch.cuda.stream(side):
                st_t = ex_tea.fwd_begin(x_tea, False)
            for bi in range(len(ex_stu.blocks)):
                ex_stu.fwd_block(st_s, bi)
                with torch.cuda.stream(side):
                    ex_tea.fwd_block(st_t, bi)
            logits_s, saved = ex_stu.fwd_end(st_s)
            with torch.cuda.stream(side):
                logits_t, _ = ex_tea.fwd_end(st_t)
        ctx.executor = ex_stu
        ctx.saved_acts = saved
        ctx.mark_non_differentiable(logits_t)
        return logits_s, logits_t

    @staticmethod
    def backward(ctx, dlogits, _unused):
        dx = ctx.executor.backward(ctx.saved_acts, dlogits.contiguous().float())
        ctx.saved_acts = None
        return dx, None, None, None, None, None


def run_body_pair(ex_stu, x_stu, ex_tea, x_tea, side):
    """(student logits [differentiable], teacher logits). `x_tea` must have been produced on `side`; the caller joins
    the streams before it reads the teacher logits on the current stream."""
    if len(ex_stu.blocks) != len(ex_tea.blocks):
        raise ValueError('student and teacher bodies differ')
    need_grad = torch.is_grad_enabled() and ex_stu.trainable
    if need_grad and not x_stu.requires_grad:
        x_stu = x_stu.detach().requires_grad_(True)
    return _BodyPairFn.apply(x_stu, x_tea, ex_stu, ex_tea, side, need_grad)


def run_body(executor, x_nhwc):
    # a gradient is needed for the weights (a trainable network) or for the INPUT alone (VAT direction through a frozen
    # teacher: torch.autograd.grad wrt the perturbation)
    need_grad = torch.is_grad_enabled() and (executor.trainable or x_nhwc.requires_grad)
    if need_grad and not x_nhwc.requires_grad:
        x_nhwc = x_nhwc.detach().requires_grad_(True)     # make sure autograd calls our backward
    return _BodyFn.apply(x_nhwc, executor, need_grad)
