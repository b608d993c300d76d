"""Import-surface shim (oracle only): attribute-access dict used by the reference's model outputs."""


class EasyDict(dict):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v
