class Data:
    def __init__(self, x=None, pos=None, **kw):
        self.x = x
        self.pos = pos

    def to(self, device):
        return self


class Batch:
    def __init__(self, data_list):
        self.data_list = data_list

    @classmethod
    def from_data_list(cls, data_list):
        return cls(data_list)

    def to(self, device):
        return self
