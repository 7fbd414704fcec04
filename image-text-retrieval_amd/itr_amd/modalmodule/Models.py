"""Model wrappers with the reference's interface (itr/modalmodule/Models.py): attributes config, img_enc,
txt_enc, sim_enc, criterion, optimizer, Eiters, logger; methods forward_emb / forward_loss / train_emb / val_start /
train_start / state_dict / load_state_dict.  Towers and losses run on the HIP kernels.

`train_emb` (forward -> loss -> backward -> clip_grad_norm_ -> Adam, Models.py:115-145, :198-225, :444-464) is built for the
GRU family with a pooled or SCAN similarity (VSE++, SCAN) and for SAEM (frozen BERT in training mode, cnn / pooling / trans
head and the image transformer layer on the tape, live dropout) and CAMERA (AGSA with BatchNorm batch statistics,
dilated-convolution summarisation, multi-view matching): itr_amd/autograd.py wires the HIP forward / backward kernels into
torch's tape.  SGRAF trains too: its similarity module runs the reference's per-caption structure on the tape in training
mode (Fusionmodule.encoder_similarity_train) and the fused kernels in evaluation mode; VSRN trains with its captioning branch.

Data-parallel training (the reference has none): with torch.distributed initialised and world > 1,
every rank receives the SAME global batch (loaders share the seed), keeps the strided shard rank::world of it, runs
the towers on the shard, all-gathers the caption (word) embeddings, scores its image rows against ALL captions,
all-gathers the score rows and evaluates the hinge on the full B x B matrix -- so the hardest negatives are searched
in the GLOBAL batch and the step equals the single-GPU one up to fp32 summation order.  Gradients: caption
embeddings sum-all-reduced inside the gather's backward, parameters in one flat bucket inside Adam.step."""
import numpy as np
import torch
from torch import nn

from .. import autograd as ag
from .. import ops
from . import ImgEncoder, TextEncoder, Objectives, Fusionmodule


class base_module(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.grad_clip = config.get('grad_clip', 2.)
        self.Eiters = 0
        self.img_enc = None
        self.txt_enc = None
        self.sim_enc = None
        self.criterion = None
        self.optimizer = None
        self.params = None
        self.logger = None

    def calculate_params(self):
        self.params_num = sum(p.numel() for p in self.params)
        if self.optimizer is None:       # every wrapper owns Adam(params, lr) like the reference (Models.py:98, :178)
            self.optimizer = ag.Adam(self.params, lr=self.config['learning_rate'])

    def state_dict(self):
        """List layout of the reference (Models.py:37-40).  The reference stores the sim_enc MODULE as the third
        entry; its state_dict is stored here, and load_state_dict accepts either."""
        sd = [self.img_enc.state_dict(), self.txt_enc.state_dict()]
        if self.sim_enc is not None:
            sd.append(self.sim_enc.state_dict())
        return sd

    @staticmethod
    def _strip_dp(sd):
        """The reference wraps CAMERA's towers in nn.DataParallel (Models.py:561-562), so its checkpoints name their tensors
        'module.<name>'; accepted for every model."""
        if isinstance(sd, dict) and sd and all(k.startswith('module.') for k in sd):
            return type(sd)((k[len('module.'):], v) for k, v in sd.items())
        return sd

    def load_state_dict(self, state_dict):
        self.img_enc.load_state_dict(self._strip_dp(state_dict[0]))
        self.txt_enc.load_state_dict(self._strip_dp(state_dict[1]))
        if self.sim_enc is not None:
            third = state_dict[2]
            self.sim_enc.load_state_dict(third.state_dict() if isinstance(third, nn.Module) else third)

    def train_start(self):
        self.img_enc.train()
        self.txt_enc.train()
        if self.sim_enc is not None:
            self.sim_enc.train()

    def val_start(self):
        self.img_enc.eval()
        self.txt_enc.eval()
        if self.sim_enc is not None:
            self.sim_enc.eval()

    def _log(self, k, v, n=0):
        if self.logger is not None:
            self.logger.update(k, v, n)

    def train_emb(self, train_data, *a, **k):
        raise NotImplementedError("%s does not define train_emb" % type(self).__name__)      # (every model family of the reference does)

    # ---- data-parallel sharding of one global batch
    def _dp_comm(self):
        from ..evalpipe import Comm
        comm = Comm()
        if self.optimizer is not None:
            self.optimizer.comm = comm if comm.on else None
        return comm

    @staticmethod
    def _dp_shard(comm, images, captions, lengths):
        """Strided shard rank::world of a length-sorted batch (every shard stays sorted and gets the same mix of
        lengths).  Returns the shard + the rank-major global bookkeeping every rank can derive without communication:
        rows per rank, lengths of all captions in gathered order."""
        B = len(lengths)
        if B < comm.world:
            raise ValueError("data-parallel train_emb: batch of %d on %d ranks" % (B, comm.world))
        sel = [list(range(q, B, comm.world)) for q in range(comm.world)]
        mine = sel[comm.rank]
        idx = torch.as_tensor(mine, dtype=torch.long)
        lens_all = [int(lengths[i]) for q in range(comm.world) for i in sel[q]]
        rows = [len(x) for x in sel]
        toks = [sum(int(lengths[i]) for i in x) for x in sel]
        lens = [int(lengths[i]) for i in mine]
        images = images[idx.to(images.device)]
        captions = captions[idx.to(captions.device)][:, :max(lens)]
        return images, captions, lens, rows, toks, lens_all

    # ---- shared by the GRU-family training steps
    def _train_towers(self, images, captions, lengths, pooled_images, last_state):
        """Differentiable towers on one batch -> (img_emb, packed word / caption embeddings, tok_off, lens).
        Same arithmetic as forward_emb; captions arrive sorted by length (collate_fn)."""
        images = self._dev(images)
        captions = self._dev(captions)
        ie, te = self.img_enc, self.txt_enc
        img = ie.forward_train(ops.mean_mid(images) if pooled_images else images)
        if getattr(te, 'dropout_p', 0.) > 0 and te.training and not hasattr(self, '_seeds'):
            self._seeds = ag.DropoutSeeds()
            self._seeds.new_step()
        if last_state != (te.method_name in ('VSE++', 'VSRN')):
            raise ValueError("_train_towers: last_state=%r does not match the text tower's pooling (method_name=%r)" % (last_state, te.method_name))
        seq, off, lens, _ = te.forward_packed_train(captions, lengths, seeds=getattr(self, '_seeds', None))
        return img, seq, off, lens

    def _step(self, loss, batch_size, logged=None):
        """backward -> clip_grad_norm_(params, grad_clip) -> Adam (Models.py:139-144, :220-225).  `logged`: the value of the
        whole batch when `loss` holds only this rank's part of a term (data parallel)."""
        self._log('Loss', loss.detach() if logged is None else logged, batch_size)      # a device scalar like the reference's loss.data: no host sync here
        loss.backward()
        self.optimizer.step(max_norm=self.grad_clip if self.grad_clip > 0 else 0.0)

    @staticmethod
    def _dev(t):
        return t.cuda() if torch.is_tensor(t) and not t.is_cuda else t


class VSE_PP(base_module):
    """VSE++ (Models.py:63-145) with the build decisions of SURVEY Q3 for *_precomp data: mean-pooled regions
    -> fc -> l2norm; text = last valid GRU state -> l2norm; honours bi_gru."""

    def __init__(self, config):
        super().__init__(config)
        if not config['data_name'].endswith('_precomp'):
            raise NotImplementedError("raw-image VSE++ (EncoderImageFull: torchvision CNN) is out of scope")
        self.img_enc = ImgEncoder.EncoderImagePooledPrecomp(config['img_dim'], config['embed_size'],
                                                            no_imgnorm=config['no_imgnorm'],
                                                            precomp_enc_type='basic', use_abs=config['use_abs'])
        self.txt_enc = TextEncoder.EncoderText(config['vocab_size'], config['word_dim'], config['embed_size'],
                                               config['num_layers'], use_bi_gru=config.get('bi_gru', False),
                                               use_abs=config['use_abs'], no_txtnorm=False, method_name='VSE++')
        self.img_enc.cuda()
        self.txt_enc.cuda()
        self.criterion = Objectives.ContrastiveLoss(config=config, margin=config['margin'],
                                                    max_violation=config['max_violation'], measure=config['measure'])
        self.params = list(self.txt_enc.parameters()) + list(self.img_enc.fc.parameters())
        self.optimizer = ag.Adam(self.params, lr=config['learning_rate'])
        self.calculate_params()

    def forward_emb(self, images, captions, lengths, *args, **kwargs):
        img_emb = self.img_enc(self._dev(images))
        cap_emb, _ = self.txt_enc(self._dev(captions), lengths)
        return img_emb, cap_emb

    def train_emb(self, train_data, *args, **kwargs):
        """One training step (Models.py:115-145)."""
        images, _, _, captions, lengths, _, _, _ = train_data
        if self.config['measure'] not in ('cosine', 'order'):
            raise ValueError("unknown measure:", self.config['measure'])
        pair_scores = ag.order_scores if self.config['measure'] == 'order' else ag.cosine_scores
        self.Eiters += 1
        self._log('Eit', self.Eiters)
        self._log('lr', self.optimizer.param_groups[0]['lr'])
        self.optimizer.zero_grad()
        comm = self._dp_comm()
        if comm.on:
            images, captions, lengths, rows, _, _ = self._dp_shard(comm, images, captions, lengths)
        with torch.enable_grad():
            img, cap, _, _ = self._train_towers(images, captions, lengths, pooled_images=True, last_state=True)
            if comm.on:
                cap = ag.dp_gather_rows(cap, comm, rows, reduce=True)
                scores = ag.dp_gather_rows(pair_scores(img, cap), comm, rows, reduce=False)
            else:
                scores = pair_scores(img, cap)
            loss = ops.hinge_loss(scores, self.config['margin'], self.config['max_violation'])
            self._step(loss, scores.size(0))

    def forward_loss(self, img_emb, cap_emb):
        loss = self.criterion(img_emb, cap_emb)
        self._log('Loss', loss.data, img_emb.size(0))
        return loss


class SCAN(base_module):
    """Stacked Cross Attention Network (Models.py:148-225)."""

    def __init__(self, config):
        super().__init__(config)
        self.img_enc = ImgEncoder.EncoderImagePrecomp(config['img_dim'], config['embed_size'],
                                                      precomp_enc_type=config['precomp_enc_type'],
                                                      no_imgnorm=config['no_imgnorm'])
        self.txt_enc = TextEncoder.EncoderText(config['vocab_size'], config['word_dim'], config['embed_size'],
                                               config['num_layers'], use_bi_gru=config['bi_gru'],
                                               no_txtnorm=config['no_txtnorm'])
        self.img_enc.cuda()
        self.txt_enc.cuda()
        self.criterion = Objectives.ContrastiveLoss(config=config, margin=config['margin'],
                                                    measure=config['measure'],
                                                    max_violation=config['max_violation'])
        self.params = list(self.txt_enc.parameters()) + list(self.img_enc.fc.parameters())
        self.optimizer = ag.Adam(self.params, lr=config['learning_rate'])
        self.calculate_params()

    def forward_emb(self, images, captions, lengths, *args, **kwargs):
        img_emb = self.img_enc(self._dev(images))
        cap_emb, cap_lens = self.txt_enc(self._dev(captions), lengths)
        return img_emb, cap_emb, cap_lens

    def train_emb(self, train_data, *args, **kwargs):
        """One training step (Models.py:198-225)."""
        images, _, _, captions, lengths, _, _, _ = train_data
        cfg = self.config
        if cfg['cross_attn'] not in ('t2i', 'i2t'):
            raise ValueError("unknown first norm type:", cfg['raw_feature_norm'])
        self.Eiters += 1
        self._log('Eit', self.Eiters)
        self._log('lr', self.optimizer.param_groups[0]['lr'])
        self.optimizer.zero_grad()
        comm = self._dp_comm()
        if comm.on:
            images, captions, lengths, rows, toks, lens_all = self._dp_shard(comm, images, captions, lengths)
        with torch.enable_grad():
            img, words, off, lens = self._train_towers(images, captions, lengths, pooled_images=False, last_state=False)
            if comm.on:
                words = ag.dp_gather_rows(words, comm, toks, reduce=True)
                lens = lens_all
                off = np.concatenate([[0], np.cumsum(lens_all)[:-1]]).astype(np.int64)
            score_fn = ag.scan_t2i_scores if cfg['cross_attn'] == 't2i' else ag.scan_i2t_scores
            scores = score_fn(img, words, off, lens, cfg['raw_feature_norm'], cfg['agg_func'], cfg['lambda_lse'], cfg['lambda_softmax'])
            if comm.on:
                scores = ag.dp_gather_rows(scores, comm, rows, reduce=False)
            loss = ops.hinge_loss(scores, cfg['margin'], cfg['max_violation'])
            self._step(loss, scores.size(0))

    def forward_loss(self, img_emb, cap_emb, cap_lens):
        loss = self.criterion(img_emb, cap_emb, cap_lens)
        self._log('Loss', loss.data, img_emb.size(0))
        return loss


class VSRN(base_module):
    """Visual Semantic Reasoning Network (Models.py:229-365): region-relationship GCN + region GRU image tower, last-state GRU text
    tower, cosine similarity, plus the training-only captioning branch (EncoderRNN / attention DecoderRNN over the GCN region
    features, LanguageModelCriterion).  The reference's checkpoints hold [img_enc, txt_enc] only (Models.py:37-45: the captioning
    model is never saved) -- the same here."""

    def __init__(self, config, use_txt_emb=True):
        super().__init__(config)
        if not config['data_name'].endswith('_precomp'):
            raise NotImplementedError("raw-image VSRN (EncoderImageFull: torchvision CNN) is out of scope")
        if use_txt_emb:
            self.img_enc = ImgEncoder.EncoderImagePrecompAttn(config['img_dim'], config['embed_size'], config['data_name'],
                                                              use_abs=config['use_abs'], no_imgnorm=config['no_imgnorm'])
        else:
            self.img_enc = ImgEncoder.EncoderImagePrecomp(config['img_dim'], config['embed_size'], use_abs=config['use_abs'],
                                                          no_imgnorm=config['no_imgnorm'])
        self.txt_enc = TextEncoder.EncoderText(config['vocab_size'], config['word_dim'], config['embed_size'],
                                               config['num_layers'], use_abs=config['use_abs'],
                                               no_txtnorm=config['no_txtnorm'], method_name=config['name'])
        self.encoder = Fusionmodule.EncoderRNN(config['dim_vid'], config['dim_hidden'], bidirectional=config['bidirectional'],
                                               input_dropout_p=config['input_dropout_p'], rnn_cell=config['rnn_type'],
                                               rnn_dropout_p=config['rnn_dropout_p'])
        self.decoder = Fusionmodule.DecoderRNN(config['vocab_size'], config['max_len'], config['dim_hidden'], config['dim_word'],
                                               input_dropout_p=config['input_dropout_p'], rnn_cell=config['rnn_type'],
                                               rnn_dropout_p=config['rnn_dropout_p'], bidirectional=config['bidirectional'])
        self.caption_model = Fusionmodule.S2VTAttModel(self.encoder, self.decoder)
        self.img_enc.cuda()
        self.txt_enc.cuda()
        self.caption_model.cuda()
        self.criterion = Objectives.ContrastiveLoss(config=config, margin=config['margin'], measure=config['measure'],
                                                    max_violation=config['max_violation'])
        self.params = list(self.txt_enc.parameters()) + list(self.img_enc.parameters()) + list(self.caption_model.parameters())
        self.calculate_params()

    def train_start(self):
        super().train_start()
        self.caption_model.train()

    def val_start(self):
        super().val_start()
        self.caption_model.eval()

    def forward_emb(self, images, captions, lengths, *args, **kwargs):
        out = self.img_enc(self._dev(images))
        img_emb, gcn_emb = out if isinstance(out, tuple) else (out, None)
        cap_emb, _ = self.txt_enc(self._dev(captions), lengths)
        return img_emb, cap_emb, gcn_emb

    def forward_loss(self, img_emb, cap_emb, GCN_img_emd=None, captions=None, captions_mask=None):
        """Retrieval loss (Models.py:337) on evaluation-mode embeddings.  The captioning term needs the autograd tape through the
        towers: it is part of train_emb (passing captions here raises)."""
        if captions is not None:
            raise NotImplementedError("the captioning loss is evaluated inside train_emb (the towers must be on the autograd tape)")
        loss = self.criterion(img_emb, cap_emb)
        self._log('Loss_retrieval', loss.data, img_emb.size(0))
        self._log('Loss', loss.data, img_emb.size(0))
        return loss

    def train_emb(self, train_data, *args, **kwargs):
        """One training step (Models.py:343-365): towers on the autograd tape -> retrieval hinge + captioning loss
        (calcualte_caption_loss :303-313) -> backward, clip_grad_norm_, Adam."""
        images, _, _, captions, lengths, _, captions_mask, _ = train_data
        if not isinstance(self.img_enc, ImgEncoder.EncoderImagePrecompAttn):
            raise NotImplementedError("VSRN.train_emb with use_txt_emb=False")
        if self.config['measure'] not in ('cosine', 'order'):
            raise ValueError("unknown measure:", self.config['measure'])
        self.Eiters += 1
        self._log('Eit', self.Eiters)
        self._log('lr', self.optimizer.param_groups[0]['lr'])
        if not hasattr(self, '_seeds'):
            self._seeds = ag.DropoutSeeds()
        self._seeds.new_step()
        comm = self._dp_comm()
        n_batch = len(lengths)
        if comm.on:
            # data parallel: a strided shard through the towers (the BatchNorms of the GCN and of the image tower take their
            # statistics over the rows of all ranks, ag.bn_sync), embeddings all-gathered, retrieval hinge replicated on the full
            # batch; the captioning loss is a sum over rows / batch size, so every rank adds its shard's part of it
            if n_batch < comm.world:
                raise ValueError("data-parallel train_emb: batch of %d on %d ranks" % (n_batch, comm.world))
            sel = torch.arange(comm.rank, n_batch, comm.world)
            rows = [len(range(q, n_batch, comm.world)) for q in range(comm.world)]
            images, captions, captions_mask = (t[sel.to(t.device)] for t in (images, captions, captions_mask))
            lengths = [lengths[i] for i in sel.tolist()]
            self._seeds.base += comm.rank * 7919                                                # other masks on other shards
        self.optimizer.zero_grad()
        with torch.enable_grad(), ag.bn_sync(comm):
            images, captions = self._dev(images), self._dev(captions)
            img, gcn_emb = self.img_enc.forward_train(images)
            te = self.txt_enc
            toks, off, lens, _ = TextEncoder.pack_tokens(captions, lengths)
            seq = ag.gru_sequence(toks, off, lens, te.embed.weight, dict(te.rnn.named_parameters()), te.use_bi_gru)
            cap = ag.gather_rows(seq, off + ops.h2d(np.asarray(lens, np.int64), off.device) - 1)
            if not te.no_txtnorm:
                cap = ag.l2norm_rows(cap)
            if te.use_abs:
                cap = cap.abs()
            if comm.on:
                img = ag.dp_gather_rows(img, comm, rows, reduce=False)
                cap = ag.dp_gather_rows(cap, comm, rows, reduce=False)
            scores = (ag.order_scores if self.config['measure'] == 'order' else ag.cosine_scores)(img, cap)
            retrieval_loss = ops.hinge_loss(scores, self.config['margin'], self.config['max_violation'])
            caption_loss = self.caption_model.caption_loss_train(gcn_emb, captions, self._dev(captions_mask), self._seeds,
                                                                 self.caption_model.training, batch_total=n_batch)
            caption_total = caption_loss.detach()
            if comm.on:
                import torch.distributed as dist
                caption_total = comm.all_reduce(caption_total.clone(), dist.ReduceOp.SUM)          # for the log only
            self._log('Loss_caption', caption_total, img.size(0))
            self._log('Loss_retrieval', retrieval_loss.detach(), img.size(0))
            self._step(retrieval_loss + caption_loss, img.size(0), logged=retrieval_loss.detach() + caption_total)


class SGRAF(base_module):
    """Similarity Reasoning and Filtration network (Models.py:468-546)."""

    def __init__(self, config):
        super().__init__(config)
        self.img_enc = ImgEncoder.EncoderImagePrecomp(config['img_dim'], config['embed_size'],
                                                      no_imgnorm=config['no_imgnorm'], precomp_enc_type='basic')
        self.txt_enc = TextEncoder.EncoderText(config['vocab_size'], config['word_dim'], config['embed_size'],
                                               config['num_layers'], use_bi_gru=config['bi_gru'],
                                               no_txtnorm=config['no_txtnorm'], dropout=.4)
        self.sim_enc = Fusionmodule.EncoderSimilarity(config['embed_size'], config['sim_dim'], config['module_name'],
                                                      config['sgr_step'])
        self.img_enc.cuda()
        self.txt_enc.cuda()
        self.sim_enc.cuda()
        self.criterion = Objectives.ContrastiveLoss(config=config, margin=config['margin'], measure=config['measure'],
                                                    max_violation=config['max_violation'])
        self.params = list(self.txt_enc.parameters()) + list(self.img_enc.parameters()) + list(self.sim_enc.parameters())
        self.calculate_params()

    def forward_emb(self, images, captions, lengths, *args, **kwargs):
        img_embs = self.img_enc(self._dev(images))
        cap_embs, _ = self.txt_enc(self._dev(captions), lengths)
        return img_embs, cap_embs

    def forward_loss(self, sims):
        loss = self.criterion(sims)
        self._log('Loss', loss.item(), sims.size(0))
        return loss

    def train_emb(self, train_data, *args, **kwargs):
        """One training step (Models.py:524-546): towers and the similarity module on the autograd tape (word-embedding and
        self-attention dropout live, BatchNorm batch statistics), hinge on the B x B similarity matrix, backward,
        clip_grad_norm_, Adam.  The similarity module follows the reference's per-caption structure in training mode
        (Fusionmodule.encoder_similarity_train); evaluation uses the fused kernels."""
        images, _, _, captions, lengths, _, _, _ = train_data
        self.Eiters += 1
        self._log('Eit', self.Eiters)
        self._log('lr', self.optimizer.param_groups[0]['lr'])
        if not hasattr(self, '_seeds'):
            self._seeds = ag.DropoutSeeds()
        self._seeds.new_step()
        self.optimizer.zero_grad()
        comm = self._dp_comm()
        shared = None
        if comm.on:
            # data parallel by CAPTION (the similarity module works caption by caption, each against all images): towers on a
            # strided shard, region embeddings all-gathered (every rank holds a partial gradient for all of them: summed in the
            # gather's backward), this rank's columns of the B x B matrix, columns all-gathered, hinge replicated.  The global
            # image vectors (VisualSA: BatchNorm over the image batch) are computed replicated on the gathered embeddings with the
            # same dropout masks everywhere; all other dropout sites draw other masks on other shards.
            images, captions, lengths, rows, _, _ = self._dp_shard(comm, images, captions, lengths)
            shared = ag.DropoutSeeds()
            shared.base, shared.n = self._seeds.base, 2048
            self._seeds.base += comm.rank * 7919
        with torch.enable_grad():
            img, words, off, lens = self._train_towers(images, captions, lengths, pooled_images=False, last_state=False)
            if comm.on:
                img = ag.dp_gather_rows(img, comm, rows, reduce=True)
            off_host = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
            sims = Fusionmodule.encoder_similarity_train(self.sim_enc, img, words, off_host, [int(x) for x in lens], self._seeds,
                                                         self.sim_enc.training, seeds_global=shared)
            if comm.on:
                sims = ag.dp_gather_rows(sims.t().contiguous(), comm, rows, reduce=False).t()      # (images, captions), both rank-major
            loss = ops.hinge_loss(sims, self.config['margin'], self.config['max_violation'])
            self._step(loss, sims.size(0))


class SAEM(base_module):
    """SAEM (Models.py:369-464): BERT text tower + transformer image tower, cosine via pdist_cos."""

    def __init__(self, config):
        super().__init__(config)
        self.img_enc = ImgEncoder.TransformerMapping(config)
        self.txt_enc = TextEncoder.BertMapping(config)
        self.txt_enc.cuda()
        self.img_enc.cuda()
        self.criterion = Objectives.ContrastiveLoss(config=config, margin=config['margin'], measure=config['measure'],
                                                    max_violation=config['max_violation'])
        self.criterion_2 = Objectives.AngularLoss()
        self.params = list(self.txt_enc.parameters()) + list(self.img_enc.parameters())
        self.calculate_params()
        # Adam over the parameters that receive gradients (the reference hands the frozen BERT weights to Adam as well; they
        # never get a gradient there either)
        self.optimizer = ag.Adam([p_ for p_ in self.params if p_.requires_grad], lr=config['learning_rate'])
        self.no_decay = ['bias', 'gamma', 'beta']

    def forward_emb(self, images, captions, captions_mask, captions_type_ids, lengths, *args, **kwargs):
        cap_embs = self.txt_enc(self._dev(captions), self._dev(captions_mask), self._dev(captions_type_ids), lengths)
        img_embs = self.img_enc(self._dev(images))
        return img_embs, cap_embs

    def forward_loss(self, epoch, img_emb, cap_emb, cap_len, ids):
        """loss1 + alpha(epoch) * AngularLoss + 0.01 * sum ||W|| over the image tower's decayed parameters
        (Models.py:419-442).  The ranking term runs on the HIP kernels; the two auxiliaries are training-time scalars
        composed of torch GPU ops."""
        alpha = 0 if epoch > 20 else 0.5 * (0.1 ** (epoch // 5))
        loss1 = self.criterion(img_emb, cap_emb, cap_len)
        loss2 = self.criterion_2(img_emb, cap_emb, cap_len, ids)
        self._log('Loss1', loss1.item(), img_emb.size(0))
        self._log('Loss2', loss2.item(), img_emb.size(0))
        l2_reg = torch.zeros((), device=img_emb.device)
        for name, param in self.img_enc.named_parameters():
            if name.split('.')[-1] not in self.no_decay:
                l2_reg = l2_reg + torch.norm(param.detach())
        return loss1 + alpha * loss2 + 0.01 * l2_reg

    def train_emb(self, train_data, epoch=0):
        """One training step (Models.py:444-464): towers on the autograd tape (frozen BERT forward in training mode, head
        and image transformer layer differentiable), loss1 + alpha(epoch) * AngularLoss + 0.01 * sum ||W||, backward,
        clip_grad_norm_, Adam.  utils.train_step never passes the epoch, so alpha is 0.5 there, as in the reference."""
        images, _, _, captions, lengths, ids, captions_mask, captions_type_ids = train_data
        self.Eiters += 1
        self._log('Eit', self.Eiters)
        self._log('lr', self.optimizer.param_groups[0]['lr'])
        if not hasattr(self, '_seeds'):
            self._seeds = ag.DropoutSeeds()
        self._seeds.new_step()
        comm = self._dp_comm()
        rows = None
        if comm.on:
            # data parallel: strided shard of the global batch through the towers, both embedding sets all-gathered; every
            # loss term is then evaluated replicated on the full batch, so the gathers' backward is a plain slice and only the
            # parameter gradients are summed (Adam.step).  The regulariser depends on the parameters alone: 1 / world of it
            # per rank.
            B = len(lengths)
            if B < comm.world:
                raise ValueError("data-parallel train_emb: batch of %d on %d ranks" % (B, comm.world))
            sel = torch.arange(comm.rank, B, comm.world)
            rows = [len(range(q, B, comm.world)) for q in range(comm.world)]
            order = [i for q in range(comm.world) for i in range(q, B, comm.world)]          # rank-major global order
            images, captions, captions_mask, captions_type_ids = (t[sel.to(t.device)] for t in (images, captions, captions_mask, captions_type_ids))
            lengths_all, ids_all = [lengths[i] for i in order], [ids[i] for i in order]
            lengths = [lengths[i] for i in sel.tolist()]
            self._seeds.base += comm.rank * 7919                                                # other masks on other shards
        self.optimizer.zero_grad()
        with torch.enable_grad():
            cap = self.txt_enc.forward_train(self._dev(captions), self._dev(captions_mask), self._dev(captions_type_ids), lengths, self._seeds)
            img = self.img_enc.forward_train(self._dev(images), self._seeds)
            if comm.on:
                img = ag.dp_gather_rows(img, comm, rows, reduce=False)
                cap = ag.dp_gather_rows(cap, comm, rows, reduce=False)
                lengths, ids = lengths_all, ids_all
            # criterion = hinge on pdist_cos (Objectives.py:310-323: rows renormalised, no eps), on the tape
            scores = ag.cosine_scores(ag.l2norm_rows(img, eps=0.0), ag.l2norm_rows(cap, eps=0.0))
            loss1 = ops.hinge_loss(scores, self.config['margin'], self.config['max_violation'])
            alpha = 0 if epoch > 20 else 0.5 * (0.1 ** (epoch // 5))
            loss2 = self.criterion_2(img, cap, lengths, ids)
            l2_reg = torch.zeros((), device=img.device)
            for name, param in self.img_enc.named_parameters():
                if name.split('.')[-1] not in self.no_decay:
                    l2_reg = l2_reg + torch.norm(param)
            loss = loss1 + alpha * loss2 + (0.01 / comm.world) * l2_reg
            self._log('Loss1', loss1.detach(), img.size(0))
            self._log('Loss2', loss2.detach(), img.size(0))
            self._step(loss, img.size(0))


class CAMERA(base_module):
    """CAMERA (Models.py:550-645): multi-view image embeddings (k x D per image), BERT + AGSA caption embedding,
    MultiViewMatching similarity, TripletLoss."""

    def __init__(self, config):
        super().__init__(config)
        self.img_enc = ImgEncoder.EncoderImagePrecompSelfAttn(config['img_dim'], config['embed_size'], config['head'],
                                                              config['smry_k'], drop=config['drop'])
        self.txt_enc = TextEncoder.CAMERAEncoderText(config['bert_config_file'], config['init_checkpoint'],
                                                     config['embed_size'], config['head'], drop=config['drop'])
        self.mvm = Fusionmodule.MultiViewMatching()
        self.img_enc.cuda()       # one process per GPU: no nn.DataParallel wrapper (Models.py:561-562)
        self.txt_enc.cuda()
        self.crit_ranking = Objectives.TripletLoss(margin=config['margin'], max_violation=config['max_violation'])
        self.crit_div = Objectives.DiversityRegularization(config['smry_k'], config['batch_size'])
        self.params = list(self.txt_enc.parameters()) + list(self.img_enc.parameters())
        self.calculate_params()
        self.optimizer = ag.Adam([p_ for p_ in self.params if p_.requires_grad], lr=config['learning_rate'])   # (frozen BERT: no gradients)

    def state_dict(self):
        """[img_enc, txt_enc] with the 'module.' prefix of the reference's nn.DataParallel wrappers (Models.py:561-562), so that a
        checkpoint written here loads in the reference on a GPU host and vice versa (load_state_dict accepts both spellings)."""
        from collections import OrderedDict
        return [OrderedDict(('module.' + k, v) for k, v in m.state_dict().items()) for m in (self.img_enc, self.txt_enc)]

    def forward_emb(self, images, boxes, imgs_wh, captions, captions_mask, captions_type_ids, *args, **kwargs):
        cap_emb = self.txt_enc(self._dev(captions), self._dev(captions_mask), self._dev(captions_type_ids))
        img_emb, smry_mat = self.img_enc(self._dev(images), self._dev(boxes), self._dev(imgs_wh))
        return img_emb, cap_emb, smry_mat

    def forward_loss(self, sim_mat, smry_mat):
        """ranking loss + smry_lamda * diversity regulariser (Models.py:598-611)."""
        ranking_loss = self.crit_ranking(sim_mat)
        self._log('Rank_Loss', ranking_loss.item(), len(sim_mat))
        div_reg = self.crit_div(smry_mat)
        self._log('Div_loss', div_reg.item(), len(sim_mat))
        loss = ranking_loss + div_reg * self.config['smry_lamda']
        self._log('Loss', loss.item(), len(sim_mat))
        return loss

    def train_emb(self, train_data, *args, **kwargs):
        """One training step (Models.py:613-645): towers on the autograd tape (frozen BERT forward in training mode, AGSA with
        BatchNorm batch statistics, dilated-convolution summarisation), multi-view matching, bidirectional hinge + smry_lamda *
        diversity regulariser, backward, clip_grad_norm_, Adam."""
        images, boxes, imgs_wh, captions, _, _, captions_mask, captions_type_ids = train_data
        self.Eiters += 1
        self._log('Eit', self.Eiters)
        self._log('lr', self.optimizer.param_groups[0]['lr'])
        if not hasattr(self, '_seeds'):
            self._seeds = ag.DropoutSeeds()
        self._seeds.new_step()
        comm = self._dp_comm()
        if comm.on:
            # data parallel (like SAEM): a strided shard of the global batch through the towers -- every BatchNorm inside takes
            # its statistics over the rows of ALL ranks (ag.bn_sync: one small all-reduce each way) -- then the multi-view
            # embeddings, caption embeddings and summarisation matrices are all-gathered and the loss is evaluated replicated on
            # the full batch, so the gathers' backward is a slice and only the parameter gradients are summed (Adam.step)
            B = len(captions)
            if B < comm.world:
                raise ValueError("data-parallel train_emb: batch of %d on %d ranks" % (B, comm.world))
            sel = torch.arange(comm.rank, B, comm.world)
            rows = [len(range(q, B, comm.world)) for q in range(comm.world)]
            images, boxes, imgs_wh, captions, captions_mask, captions_type_ids = (
                t[sel.to(t.device)] for t in (images, boxes, imgs_wh, captions, captions_mask, captions_type_ids))
            self._seeds.base += comm.rank * 7919                                                # other masks on other shards
        self.optimizer.zero_grad()
        with torch.enable_grad(), ag.bn_sync(comm):
            cap = self.txt_enc.forward_train(self._dev(captions), self._dev(captions_mask), self._dev(captions_type_ids), self._seeds)
            img, smry_mat = self.img_enc.forward_train(self._dev(images), self._dev(boxes), self._dev(imgs_wh), self._seeds)
            if comm.on:
                img = ag.dp_gather_rows(img, comm, rows, reduce=False)
                cap = ag.dp_gather_rows(cap, comm, rows, reduce=False)
                smry_mat = ag.dp_gather_rows(smry_mat, comm, rows, reduce=False)
            sim_mat = ag.mvm_scores(img, cap)
            ranking_loss = ops.hinge_loss(sim_mat, self.config['margin'], self.config['max_violation'])
            div_reg = self.crit_div(smry_mat)
            loss = ranking_loss + div_reg * self.config['smry_lamda']
            self._log('Rank_Loss', ranking_loss.detach(), len(sim_mat))
            self._log('Div_loss', div_reg.detach(), len(sim_mat))
            self._step(loss, len(sim_mat))
