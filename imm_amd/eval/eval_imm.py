"""Evaluation path of the reference on the HIP engine.

  evaluate()            imm/eval/eval_imm.py:25-139   forward passes in BN-eval mode (S12) over a dataset, collecting the
                                                       requested tensors ('gauss_yx', 'future_landmarks', ...)
  convert_landmarks()   scripts/test.py:37-46          landmarks from [-1,1] (y,x) to pixels, flattened per sample
  regress_landmarks()   scripts/test.py:48-57          Ridge(alpha=0, fit_intercept=bias) from the K unsupervised landmarks
                                                       to the annotated ones, fitted on the training split
  interocular_error()   scripts/test.py:59-65          mean point distance / inter-ocular distance (first two GT points)

The network forward is the training step's own kernel path (IMMModel.build(..., training_pl=False, build_loss=False) ->
IMMEngine.forward_model_only); the regression is host arithmetic like in the reference (scikit-learn's Ridge)."""
import time

import numpy as np
import torch


def evaluate(dataset_iter, net_instance, batch_size=100, random_seed=0, eval_tensors=None, eval_loss=False, verbose=False):
    """dataset_iter yields dicts with 'image', 'future_image' (NHWC float32 [0,255]) and optionally 'mask',
    'future_landmarks'; batches may be ragged at the end (each distinct batch size builds its engine once).
    Returns {tensor name: [per-batch numpy arrays]} like the reference."""
    np.random.seed(random_seed)
    results = {}
    step = 0
    for inputs in dataset_iter:
        t0 = time.time()
        _, loss, _, tensors = net_instance.build(inputs, training_pl=False, output_tensors=True, build_loss=eval_loss)
        tensors = dict(tensors)
        if eval_tensors is not None:
            tensors = {k: tensors[k] for k in eval_tensors}
        for k, v in tensors.items():
            results.setdefault(k, []).append(v.detach().float().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
        if verbose:
            dt = time.time() - t0
            print('test: step %d, loss = %.4f (%.1f examples/sec) %.3f sec/batch' % (
                step, float(loss) if (eval_loss and loss is not None) else 0.0, inputs['image'].shape[0] / dt, dt))
        step += 1
    return results


def convert_landmarks(tensors, im_size):
    """tensors: {'gauss_yx': [N,K,2] in [-1,1], 'future_landmarks': [N,L,2] pixels} -> (X [N, 2K], y [N, 2L])."""
    lm = np.asarray(tensors['gauss_yx'], dtype=np.float32)
    gt = np.asarray(tensors['future_landmarks']).astype(np.float32)
    lm = ((lm + 1) / 2.0) * np.array(im_size)
    n = lm.shape[0]
    return lm.reshape((n, -1)), gt.reshape((n, -1))


def regress_landmarks(train_tensors, test_tensors, im_size, bias=False):
    """Returns the regressed test landmarks [N_test, L, 2]."""
    import sklearn.linear_model
    x_train, y_train = convert_landmarks(train_tensors, im_size)
    x_test, _ = convert_landmarks(test_tensors, im_size)
    regr = sklearn.linear_model.Ridge(alpha=0.0, fit_intercept=bias)
    regr.fit(x_train, y_train)
    gt = np.asarray(test_tensors['future_landmarks']).astype(np.float32)
    return regr.predict(x_test).reshape(gt.shape)


def interocular_error(landmarks_gt, landmarks_regressed):
    """Mean over samples and points of ||gt - pred|| / ||gt eye 0 - gt eye 1|| (the first two annotated points)."""
    gt = np.asarray(landmarks_gt, dtype=np.float32)
    eyes = gt[:, :2, :]
    ocular = np.sqrt(np.sum((eyes[:, 0, :] - eyes[:, 1, :]) ** 2, axis=-1))
    dist = np.sqrt(np.sum((gt - np.asarray(landmarks_regressed)) ** 2, axis=-1))
    return float(np.mean(dist / ocular[:, None]))


def evaluate_regression(net_instance, train_iter, test_iter, im_size, batch_size=100, bias=False):
    """scripts/test.py:18-65 `evaluate`: unsupervised landmarks of both splits -> Ridge -> inter-ocular error."""
    def run(it):
        res = evaluate(it, net_instance, batch_size=batch_size, random_seed=0, eval_tensors=['gauss_yx', 'future_landmarks'])
        return {k: np.concatenate(v) for k, v in res.items()}
    train_t, test_t = run(train_iter), run(test_iter)
    pred = regress_landmarks(train_t, test_t, im_size, bias)
    return interocular_error(test_t['future_landmarks'], pred)
