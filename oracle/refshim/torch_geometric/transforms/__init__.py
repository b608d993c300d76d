class _Identity:
    def __init__(self, *a, **kw):
        pass

    def __call__(self, data):
        return data


class FixedPoints(_Identity):
    pass


class NormalizeScale(_Identity):
    pass


class RandomRotate(_Identity):
    pass


class Compose(_Identity):
    def __init__(self, ts):
        self.ts = ts

    def __call__(self, data):
        for t in self.ts:
            data = t(data)
        return data
