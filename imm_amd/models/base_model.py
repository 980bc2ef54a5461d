"""BaseModel — construction surface of /root/reference/imm/models/base_model.py:16-122.

The reference base class creates TF variables/ops; here it only keeps the bookkeeping the callers
touch (`num_instances`, `_avg_ops`, `_get_opts`, `get_bnorm_ops`) — the arithmetic of `_decay`
(:33-37) and `_exp_running_avg` (:39-50) lives in HIP kernels (imm_weight_decay_loss,
imm_perceptual_finalize).
"""


class BaseModel(object):
    num_instances = 0

    def __init__(self, dtype, name):
        self.dtype = dtype
        self._name = name
        self._avg_ops = []      # moving-average ops of the TF graph; nothing to run here
        self._opts = None
        self.__class__.num_instances += 1

    def _get_opts(self, training_pl):
        """base_model.py:62-69: weight decay 1e-5, init std 0.01."""
        if self._opts is None:
            self._opts = {'dtype': self.dtype, 'wd': 1e-5, 'std': 0.01, 'training_pl': training_pl}
        return self._opts

    def get_bnorm_ops(self, scope=None):
        """base_model.py:71-78 returns the grouped UPDATE_OPS.  The batch-norm moving averages are
        updated inside imm_bn_finalize during a training forward, so the returned op is a no-op."""
        return lambda: None

    def build(self, inputs, training_pl):
        raise NotImplementedError
