"""Stand-in for torch_geometric.data.Data as the reference uses it (data_loader.py:13,151-155,
model.py:25-26, main.py:206): an attribute bag whose .to(device) moves every tensor attribute
in place. Test infrastructure only."""
import torch


class Data(object):
    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        return self
