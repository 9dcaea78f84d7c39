"""The conversion step of the reference's demo (demo.ipynb cell 0): rhythm / pitch / timbre of one utterance replaced by
another speaker's, in the seven combinations R, F, U, RF, RU, FU, RFU.

Same inputs (two ``assets/demo.pkl``-style entries ``[speaker, emb f32[1,82], (mel[L,80], f0[L], L, uid)]``), same
padding / quantisation (``pad_seq_to_2`` to 192 frames, zero-padded F0 -> ``quantize_f0_numpy``), same F0 conversion
(``Generator_6`` logits -> argmax -> one-hot) and the same outputs ``[(name, mel[:len])]`` as the notebook.  What differs
is the schedule: the notebook runs seven batch-1 forwards, here the seven conditions are ONE batch-7 forward of the HIP
engine (every operator on the path is per-utterance, so the rows are the notebook's results).
The vocoder cell (WaveNet, external checkpoint) is out of scope.
"""
import numpy as np
import torch

from .utils import pad_seq_to_2, quantize_f0_numpy

CONDITIONS = ['R', 'F', 'U', 'RF', 'RU', 'FU', 'RFU']


def _prepare(entry, max_len_pad, device):
    mel, f0, length, uid = entry[2]
    mel_pad, _ = pad_seq_to_2(mel[np.newaxis, :, :], max_len_pad)
    f0_pad = np.pad(f0, (0, max_len_pad - length), 'constant', constant_values=(0, 0))
    onehot = quantize_f0_numpy(f0_pad)[0][np.newaxis, :, :]
    return (torch.from_numpy(mel_pad.astype(np.float32)).to(device), torch.from_numpy(onehot.astype(np.float32)).to(device),
            torch.from_numpy(np.asarray(entry[1], np.float32)).to(device), int(length), uid)


def convert_f0(P, uttr_org_pad, f0_trg_onehot):
    """Generator_6 as F0 converter: logits -> argmax -> one-hot [1, T, 257]; also returns the class indices."""
    with torch.no_grad():
        f0_pred = P(uttr_org_pad, f0_trg_onehot)[0]
        idx = f0_pred.argmax(dim=-1)
        onehot = torch.nn.functional.one_hot(idx, f0_pred.shape[-1]).to(f0_pred.dtype)[None]
    return onehot, idx


def demo_conversion(G, P, sbmt_i, sbmt_j, max_len_pad=192, device='cuda:0', conditions=CONDITIONS):
    """G: Generator_3, P: Generator_6 (both in eval mode, on `device`).  Returns [(name, mel ndarray[len, 80])]."""
    x_org, oh_org, emb_org, len_org, uid_org = _prepare(sbmt_i, max_len_pad, device)
    x_trg, oh_trg, emb_trg, len_trg, _ = _prepare(sbmt_j, max_len_pad, device)
    oh_con, _ = convert_f0(P, x_org, oh_trg)
    xf_org, xf_trg = torch.cat((x_org, oh_org), -1), torch.cat((x_org, oh_con), -1)
    x_f0 = torch.cat([xf_trg if 'F' in c else xf_org for c in conditions])
    x_rh = torch.cat([x_trg if 'R' in c else x_org for c in conditions])
    emb = torch.cat([emb_trg if 'U' in c else emb_org for c in conditions])
    with torch.no_grad():
        out = G(x_f0, x_rh, emb)
    res = []
    for n, c in enumerate(conditions):
        keep = len_trg if 'R' in c else len_org
        res.append(('{}_{}_{}_{}'.format(sbmt_i[0], sbmt_j[0], uid_org, c), out[n, :keep, :].cpu().numpy()))
    return res
