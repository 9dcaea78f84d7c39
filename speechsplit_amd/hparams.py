"""Hyper-parameters of the hot path, same names and defaults as the reference's global ``hparams`` object
(reference hparams.py:7-43).  The reference uses a vendored TF ``HParams`` container; the engine only needs
attribute access, so this is a plain attribute bag with ``values()`` / ``parse()`` for the call sites in
main.py:63 (``hparams_debug_string``) and user overrides."""


class HParams(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def values(self):
        return dict(self)

    def parse(self, text):
        """``name=value,name=value`` overrides (ints / floats / bools / strings), like HParams.parse."""
        for item in filter(None, (s.strip() for s in text.split(','))):
            k, v = item.split('=', 1)
            if k not in self:
                raise ValueError(f'unknown hyper-parameter {k}')
            old = self[k]
            if isinstance(old, bool):
                self[k] = v.lower() in ('1', 'true', 'yes')
            elif isinstance(old, int):
                self[k] = int(v)
            elif isinstance(old, float):
                self[k] = float(v)
            else:
                self[k] = v
        return self

    def copy(self):
        return HParams(self)


def default_hparams(**over):
    hp = HParams(
        # model (hparams.py:9-24)
        freq=8, dim_neck=8, freq_2=8, dim_neck_2=1, freq_3=8, dim_neck_3=32,
        dim_enc=512, dim_enc_2=128, dim_enc_3=256,
        dim_freq=80, dim_spk_emb=82, dim_f0=257, dim_dec=512, len_raw=128, chs_grp=16,
        # interp (hparams.py:27-32)
        min_len_seg=19, max_len_seg=32, min_len_seq=64, max_len_seq=128, max_len_pad=192,
        # data loader (hparams.py:35-41)
        root_dir='assets/spmel', feat_dir='assets/raptf0', batch_size=16, mode='train', shuffle=True,
        num_workers=0, samplier=8)
    hp.update(over)
    return hp


hparams = default_hparams()


def hparams_debug_string():
    values = hparams.values()
    return 'Hyperparameters:\n' + '\n'.join('  %s: %s' % (n, values[n]) for n in values)
