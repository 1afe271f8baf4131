"""ORACLE (test infrastructure, never imported by the product): plain-PyTorch restatement of the LoRA arithmetic the
reference trains adapters with.

The arithmetic lives in `peft` ([3P], unpinned: requirements.txt:9, absent from this image -> PARITY UNPINNED); the
reference's call sites are models/base.py:272-297 (`peft.LoraConfig(r, lora_alpha, lora_dropout, bias='none',
target_modules=...)` + `get_peft_model`) and models/sdxl.py:431-459 (`add_adapter` on the UNet blocks and both text
encoders; trainable tensors are cast to the adapter dtype).  Published algorithm of peft's `lora.Linear.forward`:

    result = base_layer(x) + lora_B(lora_A(lora_dropout(x))) * (lora_alpha / r)

with lora_A initialised like nn.Linear (kaiming_uniform_, a = sqrt(5)), lora_B = 0, every non-adapter parameter
frozen.  Module names: `<layer>.base_layer`, `<layer>.lora_A.<adapter>`, `<layer>.lora_B.<adapter>`.
"""
import math

from torch import nn


class LoRALinearRef(nn.Module):
    def __init__(self, base_layer, rank, alpha, dropout=0.0, adapter_name='default'):
        super().__init__()
        self.base_layer = base_layer
        self.scaling = alpha / rank
        self.adapter_name = adapter_name
        self.drop = nn.Dropout(dropout) if dropout > 0 else nn.Identity()
        self.lora_A = nn.ModuleDict({adapter_name: nn.Linear(base_layer.in_features, rank, bias=False)})
        self.lora_B = nn.ModuleDict({adapter_name: nn.Linear(rank, base_layer.out_features, bias=False)})
        nn.init.kaiming_uniform_(self.lora_A[adapter_name].weight, a=math.sqrt(5))
        nn.init.zeros_(self.lora_B[adapter_name].weight)

    def forward(self, x):
        down = self.lora_A[self.adapter_name](self.drop(x))
        return self.base_layer(x) + self.lora_B[self.adapter_name](down) * self.scaling


def apply_lora_ref(root, rank, alpha, dropout=0.0, target=None):
    """Wrap every nn.Linear under `root` that `target(name, module)` accepts; freeze all other parameters."""
    sites = []
    for name, module in root.named_modules():
        for child_name, child in module.named_children():
            full = f'{name}.{child_name}' if name else child_name
            if type(child) is nn.Linear and (target is None or target(full, child)):
                sites.append((module, child_name, child, full))
    for p in root.parameters():
        p.requires_grad_(False)
    for parent, child_name, child, _ in sites:
        parent._modules[child_name] = LoRALinearRef(child, rank, alpha, dropout)
    return [s[3] for s in sites]


def merged_weight(lora):
    """W + (alpha / r) * B A -- the weight a merged checkpoint would hold (used as a second, independent check)."""
    a, b = lora.lora_A[lora.adapter_name].weight, lora.lora_B[lora.adapter_name].weight
    return lora.base_layer.weight + lora.scaling * (b @ a)
