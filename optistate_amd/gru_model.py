"""Drop-in `RNN` with the reference's surface (gru/gru_model.py:7-49): same constructor signature, same
state_dict keys (gru.weight_ih_l{k}, gru.weight_hh_l{k}, gru.bias_ih_l{k}, gru.bias_hh_l{k}, fc.weight, fc.bias),
`.to()/.eval()/.parameters()` work because nn.GRU / nn.Linear are kept as the *weight containers*.
`forward` never calls them: it runs the HIP GRU kernels through the C-ABI.  No CPU fallback.
"""
import torch
import torch.nn as nn

from .engine import default_engine, flatten_state_dict


class RNN(nn.Module):
    def __init__(self, input_size, hidden_size, num_layers, num_classes, device, evaluate=False, use_sigmoid=True):
        super().__init__()
        self.num_layers = num_layers
        self.hidden_size = hidden_size
        self.input_size = input_size
        self.num_classes = num_classes
        self.device = device
        self.gru = nn.GRU(input_size, hidden_size, num_layers, batch_first=True)   # container only
        self.fc = nn.Linear(hidden_size, num_classes)                               # container only
        self.evaluate = evaluate
        self.use_sigmoid = use_sigmoid
        self.sigmoid = nn.Sigmoid()
        self._loaded_versions = None
        self._epoch = -1
        self._engine = None

    def _sync_weights(self, dev):
        if self._engine is None or self._engine.device != dev:
            self._engine = default_engine(dev.index or 0)
            self._loaded_versions = None
        # reload when this module's parameters changed OR somebody else loaded the (per-device, shared) engine since:
        # a context holds ONE model, so two RNN instances on one GPU, or a trainer updating the flat bucket in place,
        # must not leave this module running on foreign / stale packed weights
        versions = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if versions != self._loaded_versions or self._epoch != self._engine._inval_epoch:
            # new weights: flatten once and give them a new key; while they stay unchanged, coming back to this module after
            # another one ran (an ensemble alternating on one GPU, gru_train.py:205-217) re-selects the cached packed image
            self._flat = flatten_state_dict(self.state_dict(), self.num_layers, device=dev)
            self._key = self._engine.new_gru_key()
            self._loaded_versions, self._epoch = versions, self._engine._inval_epoch
            self._engine._gru_owner = None
        if self._engine._gru_owner is not self:
            self._engine.load_gru(self._flat, self.input_size, self.hidden_size, self.num_layers, self.num_classes,
                                  self.use_sigmoid, owner=self, key=self._key)

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("optistate_amd.RNN.forward needs a tensor on the MI355X (no CPU fallback); "
                               "move the input with .to('cuda')")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .train import gru_forward_autograd        # training path: HIP forward + HIP backward
            return gru_forward_autograd(self, x)
        self._bind_engine(x.device)
        with self._engine.lock:                            # load + forward as one unit: the context holds ONE model
            self._sync_weights(x.device)
            return self._engine.gru_forward(x)

    def _bind_engine(self, dev):
        if self._engine is None or self._engine.device != dev:
            self._engine = default_engine(dev.index or 0)  # the calling thread's context on this GPU
            self._loaded_versions = None

    def forward_windows(self, rows, window):
        """All sliding windows of ONE time-ordered row stream at once: rows (N, input_size) -> (N - window + 1, num_classes), output i
        = forward(rows[i : i + window][None]) -- the reference's evaluation loop (gru/gru_test.py:138-140 builds the windows,
        :174-191 runs them one by one) without materialising the windows and with the first layer's input projection computed once
        per row (os_gru_forward_windows).  Inference only; shapes the kernel does not take fall back to materialised windows."""
        if not rows.is_cuda:
            raise RuntimeError("optistate_amd.RNN.forward_windows needs a tensor on the MI355X (no CPU fallback)")
        self._bind_engine(rows.device)
        with self._engine.lock:
            self._sync_weights(rows.device)
            if self._engine.gru_windows_supported():
                try:
                    return self._engine.gru_forward_windows(rows, window)
                except RuntimeError as e:                      # a shape the entry point declines (-4): the materialised form below
                    if "(-4)" not in str(e):
                        raise
            return self._engine.gru_forward(rows.unfold(0, window, 1).permute(0, 2, 1).contiguous())
