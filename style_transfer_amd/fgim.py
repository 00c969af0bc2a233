"""Legacy FGIM edit: the host side of fgim_attack (/root/reference/method/yelp/model.py:454-516).

The reference walks a list of nine step sizes; for each one it takes exactly ONE gradient step from the original
latent (`it` starts at 4, the loop ends at `it >= 5`), greedy-decodes the perturbed latent and prints
`epsilon:`, `| It  5 | dis model pred p |` and the text; the text of the last step size is returned.  Here the
classifier gradient (one kernel) and the 9*B decodes (one batched greedy decode) run for all sentences at once.
"""
import numpy as np
import torch

from . import ops
from .data import id2text_sentence

EPSILONS = (0.01, 0.012, 0.014, 0.015, 0.016, 0.017, 0.018, 0.019, 0.02)   # model.py:467


def fgim_attack(ae, classifier, latent, target, id_to_word, gold_ids=None, epsilons=EPSILONS, max_len=60, start_id=2,
                log=print):
    """latent [B,256] (device f32), target [B] 0/1 -> (texts [B][n_eps], pred [B]); the reference returns texts[b][-1].
    `log` receives the lines the reference prints (per sentence; batch order)."""
    pred, _, out = ops.fgim_perturb(classifier, latent, target, epsilons)
    B, n = latent.shape[0], len(epsilons)
    ids, _, _ = ops.greedy_decode(ae, out.view(n * B, 256), max_len=max_len, start_id=start_id)
    ids = ids.view(n, B, -1).cpu().numpy()
    pred = pred.cpu().numpy()
    texts = []
    for b in range(B):
        if gold_ids is not None:
            log("gold: " + id2text_sentence(gold_ids[b], id_to_word))
        row = []
        for k, e in enumerate(epsilons):
            log("epsilon: {}".format(e))
            log("| It {:2d} | dis model pred {:5.4f} |".format(5, float(pred[b])))
            row.append(id2text_sentence(ids[k, b], id_to_word))
            log(row[-1])
        texts.append(row)
    return texts, pred
