"""Engine <-> TensorFlow bundle (imm_amd/utils/tf_checkpoint.py): what a trained engine writes restores bit-identically
into a fresh engine, under the reference's variable names and restore switches (cnn_train_multi.py:404-433)."""
import numpy as np
import pytest
import torch

from oracle import imm_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _model(seed=1):
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.utils.box import Box
    return IMMModel(Box(dict(O.default_model_config(10))), dtype=torch.bfloat16, device=DEV, seed=seed)


def test_engine_round_trip_through_a_tf_bundle(tmp_path):
    from imm_amd.train.cnn_train_multi import TrainStep
    from imm_amd.utils import tf_checkpoint as T
    model = _model()
    ts = TrainStep(model, 2, 128, world_size=1, use_graph=False)
    inputs = O.synthetic_inputs(2, 128, seed=0)
    for _ in range(3):
        ts.step(inputs)
    ts.synchronize()
    eng = model.engine
    prefix = str(tmp_path / 'model.ckpt-3')
    T.save_tf_checkpoint(eng, prefix)
    listing = T.list_bundle(prefix)
    # trainable + BN moving statistics (23 layers x 2) + 6 loss normalisers + the cost moving-average shadows (one 4-vector, round 6)
    # + global_step + 2 Adam slots each + 2 beta powers
    n_train = len(eng.pview)
    assert n_train == 96 and len(eng.state) == 46
    assert len(listing) == n_train + 46 + 6 + 1 + 1 + 2 * n_train + 2 and listing[T.COST_EMA_VAR][1] == (4,)
    assert listing['model/image_encoder/encoder/conv_1/conv_1/w'][1] == (7, 7, 3, 32)
    assert listing['model/renderer/conv_8/conv_8/b/Adam_1'][1] == tuple(eng.pview['model/renderer/conv_8/b'].shape)
    assert 'SelfSupReconstructionLoss/input_agg' in listing and listing['global_step'][1] == ()
    raw = T.read_bundle(prefix, names={'global_step', 'beta1_power'})
    assert float(raw['global_step']) == 3.0 and abs(float(raw['beta1_power']) - 0.9 ** 4) < 1e-7

    fresh = _model(seed=5)._get_engine(2, 128)
    assert not torch.equal(fresh.params, eng.params)
    missing = T.load_tf_checkpoint(fresh, prefix, restore_optim=True)
    assert missing == []
    assert torch.equal(fresh.params, eng.params) and torch.equal(fresh.adam_m, eng.adam_m) and torch.equal(fresh.adam_v, eng.adam_v)
    assert int(fresh.step_count) == 3 and torch.equal(fresh.loss_agg, eng.loss_agg)
    # the `_avg` summaries continue after a resume (tf.train.Saver stores the shadow variables): three updates so far
    assert torch.equal(fresh.cost_ema, eng.cost_ema) and float(eng.cost_ema[3]) == 3.0 and float(eng.cost_ema[2]) > 0.0
    for k, v in eng.state.items():
        assert torch.equal(fresh.state[k], v), k
    # the restored engine continues exactly like the original (same forward loss on the same batch)
    eng.set_inputs(inputs['image'], inputs['future_image'], inputs['mask']); eng.forward(True)
    fresh.set_inputs(inputs['image'], inputs['future_image'], inputs['mask']); fresh.forward(True)
    torch.cuda.synchronize()
    assert float(eng.loss) == float(fresh.loss)

    # model variables only: Adam state stays at its initial zeros; the step is still restored (a model variable upstream)
    other = _model(seed=6)._get_engine(2, 128)
    T.load_tf_checkpoint(other, prefix, restore_optim=False)
    assert torch.equal(other.params, eng.params) and float(other.adam_m.abs().sum()) == 0.0 and int(other.step_count) == 3
    T.load_tf_checkpoint(other, prefix, reset_global_step=0)
    assert int(other.step_count) == 0


def test_missing_variables_follow_the_reference_switch(tmp_path):
    from imm_amd.utils import tf_checkpoint as T
    eng = _model()._get_engine(2, 128)
    tensors = T.engine_to_tf(eng, with_optimizer=False)
    dropped = 'model/renderer/conv_8/conv_8/w'
    kept = eng.pview['model/renderer/conv_8/w'].clone()
    del tensors[dropped]
    prefix = str(tmp_path / 'partial.ckpt')
    T.write_bundle(prefix, tensors)
    with pytest.raises(KeyError):
        T.load_tf_checkpoint(eng, prefix)
    assert T.load_tf_checkpoint(eng, prefix, ignore_missing_vars=True) == [dropped]
    assert torch.equal(eng.pview['model/renderer/conv_8/w'], kept)
    with pytest.raises(KeyError):                                  # Adam slots are not in a model-only bundle
        T.load_tf_checkpoint(eng, prefix, restore_optim=True)
    tensors[dropped] = np.zeros((3, 3, 32, 5), np.float32)         # wrong shape
    T.write_bundle(prefix, tensors)
    with pytest.raises(ValueError, match='shape'):
        T.load_tf_checkpoint(eng, prefix)


def test_f16_loss_scale_state_travels_with_the_optimizer_slots(tmp_path):
    """f16 storage: the dynamic loss scale S and its clean-step counter are saved as an extra (non-TensorFlow) variable and come
    back with --restore-optim, so that a resume continues at the scale the run had reached; a restore without the
    optimizer starts the scale afresh like the slots."""
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.train.cnn_train_multi import TrainStep
    from imm_amd.utils import tf_checkpoint as T
    from imm_amd.utils.box import Box

    def model(seed, ls):
        return IMMModel(Box(dict(O.default_model_config(10))), dtype=torch.float16, device=DEV, seed=seed, hparams={'loss_scale': ls})
    m = model(1, 2.0 ** 22)                       # far too large: the first steps overflow, S is halved, updates are skipped
    ts = TrainStep(m, 2, 128, world_size=1, use_graph=False)
    inputs = O.synthetic_inputs(2, 128, seed=0)
    for _ in range(8):
        ts.step(inputs)
    ts.synchronize()
    eng = m.engine
    state = eng.loss_scale_state.tolist()
    assert state[2] >= 1.0 and state[0] < 2.0 ** 22 and int(eng.step_count) == 8 - int(state[2]), state
    prefix = str(tmp_path / 'model.ckpt-f16')
    T.save_tf_checkpoint(eng, prefix)
    assert T.LOSS_SCALE_VAR in T.list_bundle(prefix)
    fresh = model(5, 4096.0)._get_engine(2, 128)
    T.load_tf_checkpoint(fresh, prefix, restore_optim=True)
    assert fresh.loss_scale_state.tolist() == state and int(fresh.step_count) == int(eng.step_count) and int(fresh.adam_t) == int(eng.adam_t)
    other = model(6, 4096.0)._get_engine(2, 128)
    T.load_tf_checkpoint(other, prefix, restore_optim=False)
    assert other.loss_scale_state.tolist()[0] == 4096.0 and int(other.adam_t) == 0
