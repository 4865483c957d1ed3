"""Drop-in for the step of ``trainer/RL_TDA.py``: ``RT_TDA_Trainer.RL_TDA_train_step`` (:110-200) and the loop body around it
(:205-226) on the HIP path.

One step is: net1 = PoseNet9D() on the cloud with gradients (tgpose_amd.autograd: HIP forward and backward), net2 =
PoseNet9D(only_encoder=True) on the augmented cloud under ``no_grad`` (the fused training-mode forward of tgpose_amd.engine:
batch-statistics BatchNorm that moves its running statistics), ``feat_consistency_loss`` + two ``prop_sym_matching_loss``
(losses/consistency_loss.py), the fourteen ``control_loss('TDA')`` terms of ``TDA_loss``, and
``total = 0.1 (con + recon_1 + recon_consistency) + 0.9 sum(TDA)`` (:214).  Nothing in it reads a value back from the device,
so forward + loss + backward replay as one hipGraph (``graphed_step``); the gradient exchange between ranks
(tgpose_amd.shard), ``clip_grad_norm_(net1, 5)`` (:223) and the optimizer step stay outside the graph.

Not rebuilt (SURVEY section 8 scope): the epoch loop, logging, checkpointing, the Ranger optimizer / flat-and-anneal schedule
(``tools/training_utils.build_optimizer`` exists only as bytecode in the reference; any ``torch.optim`` optimizer over
``net1.parameters()`` is accepted) -- the data-parallel hot path is the step.
"""
import math
import os

import torch

from ..config import FLAGS
from ..losses.TDA_loss_sym_recon import TDA_loss
from ..losses.consistency_loss import feat_consistency_loss, prop_sym_matching_loss
from ..network.fs_net_repo.PoseNet9D import PoseNet9D
from .organize_loss import control_loss

PRED_KEYS = ['recon', 'p_green_R', 'p_red_R', 'f_green_R', 'f_red_R', 'Pred_T', 'Pred_s', 'h1', 'h2']


def get_gt_v(Rs, axis=2):
    """tools/training_utils.get_gt_v (bytecode only in the reference): the ground-truth green (y) and red (x) axes"""
    return Rs[:, :, 1].contiguous(), Rs[:, :, 0].contiguous()


def create_network(mode):
    if mode == 'RL_TDA':
        return PoseNet9D(), PoseNet9D(only_encoder=True)
    raise NotImplementedError(mode)


# net2's forward on a second stream beside net1's.  Built in round 3 and left off then (18.45 ms per step against 18.25 serial); on
# the round-5 kernels the branch pays: 16.44 against 17.10 ms per step (same box, alternating runs), so it is the default.  Bit for bit
# the serial step's results (tests/test_gpu_parity.py::test_train_step_with_net2_beside_net1_equals_serial).  TGP_NET2_BESIDE=0: serial.
NET2_BESIDE = os.environ.get("TGP_NET2_BESIDE", "1") != "0"
_TOTAL_W = {}


def total_loss(loss_dict):
    """trainer/RL_TDA.py:209-214: 0.1 (RL + recon_1 + recon_consistency) + 0.9 sum over the TDA terms -- as one concatenation, one
    multiply with the constant weights and one sum (Python's `sum` over seventeen device scalars is ~35 launches forward and ~70
    backward; the value differs from the left-to-right sum by the rounding of a 17-term fp32 sum)."""
    head = [loss_dict[k].reshape(-1) for k in ('RL_loss', 'recon_1_loss', 'recon_consistency_loss') if k in loss_dict]
    tda = [v.reshape(-1) for v in loss_dict['TDA_loss'].values()]
    terms = torch.cat(head + tda)
    key = (terms.device, sum(t.numel() for t in head), terms.numel())
    if key not in _TOTAL_W:          # built once per shape (the first call of a shape is an eager warm-up, outside any capture)
        w = torch.full((key[2],), 0.9)
        w[: key[1]] = 0.1
        _TOTAL_W[key] = w.to(terms.device)
    return (terms * _TOTAL_W[key]).sum()


class RT_TDA_Trainer(object):
    def __init__(self, logger=None, device=None):
        self.logger = logger
        self.device = torch.device('cuda:0') if device is None else torch.device(device)
        self.net1, self.net2 = None, None
        self.loss_tda_net = None
        self.optimizer = None
        self.scheduler = None
        self._graphed = None          # the last GraphedStep captured over net1 (its static .grad buffers must stay in place)
        self._buckets = None          # shard.GradBuckets of a graphed_step(overlap=True): p.grad are views into its flat buffers
        self._exchanged = False       # set by a replay whose gradient exchange has already run, cleared by finish_step

    def setup(self, mode, optimizer=None, scheduler=None):
        self.init_network(mode)
        self.init_loss()
        self.optimizer, self.scheduler = optimizer, scheduler

    def init_network(self, mode):
        self.net1, self.net2 = create_network(mode)
        self.net1, self.net2 = self.net1.to(self.device), self.net2.to(self.device)

    def init_loss(self):
        self.loss_tda_net = TDA_loss()
        (self.name_fs_list, self.name_recon_list, self.name_geo_list, self.name_prop_list, self.name_TDA_list) = control_loss('TDA')

    def build_params(self, training_stage_freeze=None):
        return [{"params": filter(lambda p: p.requires_grad, self.net1.parameters()), "lr": float(FLAGS.lr) * FLAGS.lr_pose}]

    # -------------------------------------------------------------------------------------------------------------------
    def losses(self, db, results, results_2, only_TDA=False, gt_pred_flag=False):
        """the loss half of RL_TDA_train_step (:121-178) on tensors already on the device"""
        dev = self.device
        PC = db['pcl_in']
        gt_R, gt_t, sym = db['rotation'], db['translation'], db['sym_info']
        loss_dict = {}
        if not only_TDA:
            recon_1 = results['recon']
            loss_dict['RL_loss'] = feat_consistency_loss(results['feat_global'], results_2['feat_global'])
            loss_dict['recon_1_loss'] = prop_sym_matching_loss(PC, recon_1, gt_R, gt_t, sym)
            loss_dict['recon_consistency_loss'] = 0.2 * prop_sym_matching_loss(recon_1, results_2['recon'], gt_R, gt_t, sym)
        else:
            loss_dict['RL_loss'] = torch.zeros(1, device=dev)
        pred_TDA_list = {'Rot1': results['p_green_R'], 'Rot1_f': results['f_green_R'], 'Rot2': results['p_red_R'],
                         'Rot2_f': results['f_red_R'], 'Recon': results['recon'], 'Tran': results['Pred_T'], 'Size': results['Pred_s'],
                         'TDA_h1': results['h1'], 'TDA_h2': results['h2']}
        gt_green_v, gt_red_v = get_gt_v(gt_R)
        gt_TDA_list = {'Rot1': gt_green_v, 'Rot2': gt_red_v, 'Recon': PC, 'Tran': gt_t, 'Size': db['fsnet_scale'], 'h1': db['pdh1'],
                       'h2': db['pdh2'], 'proto': None, 'pdh1_category': db['pdh1_category'], 'pdh2_category': db['pdh2_category'],
                       'points_category': db['points_category'], 'R': gt_R}
        loss_dict['TDA_loss'] = self.loss_tda_net(self.name_TDA_list, pred_TDA_list, gt_TDA_list, sym, gt_pred_flag)
        return loss_dict

    def RL_TDA_train_step(self, db, only_TDA=False, gt_pred_flag=False, *, sample_idx=None, inject=None, cut=None):
        """trainer/RL_TDA.py:110-200.  db: the loader's batch dict (tensors on any device).  sample_idx: optionally the
        subsamples of the two forwards, [(pool_1, pool_2) of net1, (pool_1, pool_2) of net2] (drawn from torch's global CPU
        generator in that order otherwise, as the reference does); inject: neighbour graphs for parity tests."""
        dev = self.device
        db = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in db.items()}
        PC, obj_id = db['pcl_in'], db['cat_id']
        FLAGS.train = 1                                           # the trainer runs with FLAGS.train set (engine/train.py)
        s = sample_idx if sample_idx is not None else [None, None]
        results_2 = None
        beside = NET2_BESIDE and not only_TDA and dev.type == 'cuda'
        if beside:
            # net2 (no gradients, its own cloud) shares nothing with net1 until the losses: it runs on a second stream -- in a
            # captured step a parallel branch of the graph -- and fills the gaps between net1's chains of small launches.  The
            # subsamples are drawn up front in the reference's order (net1's two, then net2's: gcn3d.py:241-242).
            from .. import engine
            s = [x if x is not None else engine.draw_sample_idx(n) for x, n in zip(s, (PC.shape[1], db['aug_pcl_in'].shape[1]))]
            cur, side = torch.cuda.current_stream(dev), engine._side_stream(dev, ("train", "net2"))
            fork = torch.cuda.Event()
            fork.record(cur)
            with torch.cuda.stream(side):
                side.wait_event(fork)
                with torch.no_grad():
                    results_2 = self.net2(db['aug_pcl_in'], obj_id, sample_idx=s[1], inject=inject)
                join = torch.cuda.Event()
                join.record(side)
            if not torch.cuda.is_current_stream_capturing():      # (a captured graph owns its pool: nothing to protect)
                db['aug_pcl_in'].record_stream(side), obj_id.record_stream(side)
        results = self.net1(PC, obj_id, sample_idx=s[0], inject=inject, cut=cut)
        if beside:
            cur.wait_event(join)
            if not torch.cuda.is_current_stream_capturing():
                for t in results_2.values():
                    if torch.is_tensor(t):
                        t.record_stream(cur)
        elif not only_TDA:
            with torch.no_grad():
                results_2 = self.net2(db['aug_pcl_in'], obj_id, sample_idx=s[1], inject=inject)
        loss_dict = self.losses(db, results, results_2, only_TDA, gt_pred_flag)
        output_dict = {'enc_feat_1': results['feat_global'], 'PC': PC, 'obj_id': obj_id, 'gt_R': db['rotation'],
                       'gt_t': db['translation'], 'gt_s': db['fsnet_scale'], 'gt_h1': db['pdh1'], 'gt_h2': db['pdh2'], 'sem_pro': None}
        if not only_TDA:
            output_dict['enc_feat_2'] = results_2['feat_global']
        for key in PRED_KEYS:
            output_dict[key] = results[key]
        return output_dict, loss_dict

    # -------------------------------------------------------------------------------------------------------------------
    def loss_is_nan(self, total):
        """the loop's NaN test (:217-220) -- one host read of the device scalar.  Data parallel: the ranks must skip or step
        TOGETHER (a rank that skipped would miss the collective the others wait in), so the flag is max-reduced first."""
        import torch.distributed as dist
        bad = torch.isnan(total.detach()).reshape(-1).any().to(torch.int32)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            bad = bad.to(self.device) if dist.get_backend() == "nccl" else bad.cpu()
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        return bool(bad.item())

    def finish_step(self, total=None):
        """what follows total_loss.backward() in the loop (:223-226), with the data-parallel gradient exchange in front: the
        clip must see the averaged gradients.  After a graphed_step(overlap=True) replay the exchange has already
        run (bucket by bucket, overlapped with the backward): the replay says so through ``_exchanged``.

        total: the replayed step's loss (graphed_step's return value).  When given, the reference's NaN test (:217-220) is
        applied here -- a captured step has run its backward before anyone can look at the loss, so a NaN step is undone by
        zeroing the gradients instead of skipping backward(): no clip, no optimizer / scheduler step, weights untouched.
        Returns False for such a skipped step."""
        from .. import shard
        if total is not None and self.loss_is_nan(total):
            print('Found nan in total loss')
            self._exchanged = False
            grads = [p.grad for p in self.net1.parameters() if p.grad is not None]
            if grads:
                torch._foreach_zero_(grads)
            return False
        if not self._exchanged:
            shard.allreduce_gradients(self.net1.parameters())
        self._exchanged = False
        torch.nn.utils.clip_grad_norm_(self.net1.parameters(), 5)
        if self.optimizer is not None:
            self.optimizer.step()
        if self.scheduler is not None:
            self.scheduler.step()
        return True

    def train_iteration(self, db):
        """one eager iteration of RL_TDA_train's loop body (:205-226); returns (total loss, loss_dict).  A NaN total skips
        backward, clip and the optimizer step, as the reference's loop does (:217-220)."""
        if self.optimizer is not None:
            # once a captured step or flat gradient buckets exist, .grad tensors are static storage: zero them in place
            self.optimizer.zero_grad(set_to_none=self._graphed is None and self._buckets is None)
        _, loss_dict = self.RL_TDA_train_step(db)
        total = total_loss(loss_dict)
        if self.loss_is_nan(total):
            print('Found nan in total loss')
            return total.detach(), loss_dict
        total.backward()
        self._exchanged = False
        self.finish_step()
        return total.detach(), loss_dict

    def graphed_step(self, db, overlap=False, _debug=""):
        """Capture forward (both nets) + losses + backward for batches of db's shapes as hipGraphs; returns a callable
        ``step(db=None, sample_idx=None) -> total loss`` that copies a new batch into the static buffers, replays, and leaves the
        gradients in net1's ``.grad`` buffers; then call ``finish_step(total=loss)``, which applies the loop's NaN test
        (:217-220) to the returned device scalar and skips clip + optimizer for a NaN step (``finish_step()`` without the loss
        steps unconditionally and reads nothing back).

        overlap=True (the data-parallel form): the backward is captured in two segments split at the encoder's output, net1's
        gradients live in two flat buckets (shard.GradBuckets), and the exchange of the late layers' bucket (84 of 97 MB) is
        started between the segments, so it runs while the encoder's backward computes; the encoder's bucket follows.  With one
        process the exchanges are no-ops and the step computes exactly what overlap=False computes."""
        from ..autograd import GraphedStep, EncoderCut, LATE_PREFIXES
        from .. import shard
        dev = self.device
        static = {k: v.to(dev).clone() for k, v in db.items() if torch.is_tensor(v)}
        N = static['pcl_in'].shape[1]
        cut = EncoderCut() if (overlap and "nocut" not in _debug) else None
        pending = []

        def step_fn(samples):
            _, loss_dict = self.RL_TDA_train_step(static, sample_idx=samples, cut=cut)
            return total_loss(loss_dict)

        between = after = None
        if overlap and "nobuckets" not in _debug:
            self._buckets = buckets = shard.GradBuckets(self.net1.named_parameters(), LATE_PREFIXES)

            def between():
                # reduce-scatter, the shard's average and the all-gather of the late bucket are all enqueued here, on the
                # exchange stream: the whole exchange travels while the encoder's backward (graph 2) computes
                pending.append(buckets.reduce(0))

            def after():
                pending.append(buckets.reduce(1))
                while pending:
                    buckets.wait(pending.pop(0))
                self._exchanged = True                            # finish_step must not exchange again
        else:
            self._buckets = None                                  # an earlier overlap capture's buckets no longer describe this step

        g = GraphedStep(list(self.net1.parameters()), step_fn, [N, N], dev, cut=cut, between=between, after=after,
                        own_pool="ownpool" in _debug, buckets=self._buckets)
        self._graphed = g

        def step(db=None, sample_idx=None):
            if db is not None:
                for k, v in db.items():
                    if torch.is_tensor(v):
                        static[k].copy_(v.reshape(static[k].shape), non_blocking=True)
            return g(sample_idx)
        step.graph = g
        return step
