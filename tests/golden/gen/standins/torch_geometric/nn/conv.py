"""Stand-in for torch_geometric.nn.conv.MessagePassing as the reference uses it (model.py:4,47-50,
99-101): aggr='add', default flow source_to_target. Published semantics restated:
  * propagate(edge_index, size=None, **kwargs) calls self.message(...) with arguments resolved by
    NAME from kwargs; a parameter called `<k>_j` receives kwargs[k].index_select(0, edge_index[0])
    (source rows), `<k>_i` receives kwargs[k].index_select(0, edge_index[1]);
  * the message result [E, *] is scatter-added onto edge_index[1] with dim_size = number of nodes
    (size of the node-feature tensor), in edge order on CPU;
  * self.update(aggr_out) post-processes the sum.
Test infrastructure only."""
import inspect

import torch


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr='add', flow='source_to_target', **kwargs):
        super(MessagePassing, self).__init__()
        assert aggr == 'add' and flow == 'source_to_target'
        self._msg_args = [p for p in inspect.signature(self.message).parameters]

    def propagate(self, edge_index, size=None, **kwargs):
        num_nodes = None
        call = []
        for name in self._msg_args:
            if name.endswith('_j') or name.endswith('_i'):
                base = kwargs[name[:-2]]
                num_nodes = base.size(0)
                sel = edge_index[0] if name.endswith('_j') else edge_index[1]
                call.append(base.index_select(0, sel))
            elif name == 'edge_index':
                call.append(edge_index)
            else:
                call.append(kwargs.get(name))
        msg = self.message(*call)
        if size is not None:
            num_nodes = size if isinstance(size, int) else size[1]
        out = torch.zeros((num_nodes,) + tuple(msg.shape[1:]), dtype=msg.dtype, device=msg.device)
        out.index_add_(0, edge_index[1], msg)
        return self.update(out)
