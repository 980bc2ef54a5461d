"""Attribute-access dict with the `hasattr` semantics the IMM model code relies on
(/root/reference/imm/models/imm_model.py:285,349,378 test optional keys with hasattr).
The reference vendors python-box (imm/utils/box.py); only this behaviour is needed on the hot path.
"""


class Box(dict):
    def __init__(self, *args, **kwargs):
        super(Box, self).__init__(*args, **kwargs)
        for k, v in list(self.items()):
            self[k] = self._wrap(v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, Box):
            return cls(v)
        if isinstance(v, list):
            return [cls._wrap(x) for x in v]
        return v

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key)

    def __setattr__(self, key, value):
        self[key] = self._wrap(value)

    def __setitem__(self, key, value):
        super(Box, self).__setitem__(key, self._wrap(value))

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, Box) else v) for k, v in self.items()}
