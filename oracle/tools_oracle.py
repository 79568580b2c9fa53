"""Restatement of the reference's evaluation-tool PLUMBING (tools/test_*_hf.py) -- TEST INFRASTRUCTURE ONLY.

SURVEY.md §8 f3: the reference's own entry points (zero-shot classification, reconstruction metrics, linear probing) drive a
VTPModel through its public methods only.  The tools themselves need torchvision / datasets and do not travel to the GPU box, so
their model-facing functions are restated here, each citing the reference lines it follows, and pinned two ways:
  * tests/test_tools_oracle.py (CPU): against the REAL tool functions imported from /root/reference/tools (skipped where the tree
    is absent) and against tests/golden/tools_tiny.safetensors, which oracle/make_golden_tools.py wrote from the real tools
    driving the real reference VTPModel (tiny seeded weights);
  * tests/test_tools_gpu.py (GPU): the same functions drive `vtp_amd.VTPModel` / a `patch_model`-ed instance and must reproduce
    the golden outputs within the bf16 noise of the reference algorithm.
`model` is anything with the reference's method surface (modeling_vtp.py:184-472)."""
from __future__ import annotations

from itertools import islice
from typing import Callable, Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import vtp_oracle as O

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)  # tools/test_reconstruction_hf.py:41-42
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


# ------------------------------------------------------------------------------------------------ a tiny deterministic tokenizer
def toy_tokenizer(vocab_size: int, context_length: int) -> Callable[[Sequence[str]], torch.Tensor]:
    """Stand-in for vtp.tokenizers.get_tokenizer (BPE, vocab 49408) for models with a small vocabulary: SOT = vocab - 2, one id per
    word (stable polynomial hash), EOT = vocab - 1, zero padding -- the layout the argmax pooling of text_transformer.py:222-224
    relies on (EOT carries the largest id).  A tokenizer is an ARGUMENT of the tool functions, so any callable does."""
    def tok(texts: Sequence[str]) -> torch.Tensor:
        out = torch.zeros(len(texts), context_length, dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [vocab_size - 2]
            for w in t.lower().replace(".", " ").replace(",", " ").split():
                h = 0
                for ch in w:
                    h = (h * 131 + ord(ch)) % 1000003
                ids.append(1 + h % (vocab_size - 3))
            ids = ids[: context_length - 1] + [vocab_size - 1]
            out[i, : len(ids)] = torch.tensor(ids)
        return out
    return tok


CLASSNAMES = ("tench", "goldfish", "great white shark", "tiger shark", "hammerhead shark", "electric ray", "stingray")
TEMPLATES = (lambda c: f"a photo of a {c}.", lambda c: f"a blurry photo of the {c}.", lambda c: f"art of the {c}.")


# ------------------------------------------------------------------------------------------------ zero-shot (tools/test_zero_shot_hf.py)
def batched(iterable, n: int):
    """:302-309"""
    it = iter(iterable)
    while True:
        batch = list(islice(it, n))
        if not batch:
            break
        yield batch


def accuracy(output: torch.Tensor, target: torch.Tensor, topk: Tuple[int, ...] = (1,)) -> List[float]:
    """:312-316"""
    pred = output.topk(max(topk), 1, True, True)[1].t()
    correct = pred.eq(target.view(1, -1).expand_as(pred))
    return [float(correct[:k].reshape(-1).float().sum(0, keepdim=True).cpu().numpy()) for k in topk]


def build_zero_shot_classifier(model, tokenizer, classnames: Sequence[str], templates: Sequence[Callable],
                               num_classes_per_batch: int = 10, device=None) -> torch.Tensor:
    """:342-394 -- per class batch: every template's text through get_clip_text_feature(normalize=True), mean over the templates,
    re-normalise, transpose; batches concatenated to [embed_dim, num_classes]."""
    num_templates = len(templates)

    def _process_batch(batch_classnames):
        texts = [template(c) for c in batch_classnames for template in templates]
        tokens = tokenizer(texts).to(device)
        text_features = model.get_clip_text_feature(tokens, normalize=True)
        text_features = text_features.reshape(len(batch_classnames), num_templates, -1).mean(dim=1)
        text_features = F.normalize(text_features, dim=1)
        return text_features.T

    with torch.no_grad():
        return torch.cat([_process_batch(b) for b in batched(classnames, num_classes_per_batch)], dim=1)


def zero_shot_evaluate(model, classifier: torch.Tensor, batches, device) -> Tuple[float, float, torch.Tensor]:
    """:401-441 at precision 'fp32' (no autocast, fp32 inputs): logits = 100 * image_features @ classifier, top-1 / top-5 in percent;
    also returns the logits of all batches (the tool does not, the tests compare them)."""
    top1, top5, n, all_logits = 0.0, 0.0, 0, []
    with torch.inference_mode():
        for images, targets in batches:
            images = images.to(device=device, dtype=torch.float32)
            targets = targets.to(device)
            image_features = model.get_clip_image_feature(images, normalize=True)
            logits = 100.0 * image_features @ classifier
            acc1, acc5 = accuracy(logits, targets, topk=(1, 5))
            top1 += acc1
            top5 += acc5
            n += images.size(0)
            all_logits.append(logits.float().cpu())
    return top1 / n * 100, top5 / n * 100, torch.cat(all_logits)


# ------------------------------------------------------------------------------------------------ reconstruction (tools/test_reconstruction_hf.py)
def calculate_psnr(original: torch.Tensor, processed: torch.Tensor) -> float:
    """:49-63 (inputs in 0..255)"""
    mse = torch.mean((original - processed) ** 2)
    if mse == 0:
        return float("inf")
    return 20 * torch.log10(torch.tensor(255.0) / torch.sqrt(mse)).item()


def denormalize(x: torch.Tensor) -> torch.Tensor:
    """transform_rev = Normalize([-m / s], [1 / s]) (:265-268): (x - (-m / s)) / (1 / s) per channel"""
    m = torch.tensor([-m / s for m, s in zip(IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD)], device=x.device).view(1, 3, 1, 1)
    s = torch.tensor([1 / s for s in IMAGENET_DEFAULT_STD], device=x.device).view(1, 3, 1, 1)
    return (x - m) / s


def reconstruct_and_psnr(model, images: torch.Tensor, encode_dtype=torch.float32):
    """:360-376, :394-397 -- latents under autocast(encode_dtype), decode under autocast(fp32) = no autocast, de-normalise, clamp to
    [0, 1], PSNR per image on the 0..255 scale.  (The reference hard-codes device_type='cuda' in both autocast contexts.)"""
    dev = images.device.type
    with torch.no_grad():
        if encode_dtype == torch.float32:
            latents = model.get_reconstruction_latents(images)
        else:
            with torch.amp.autocast(device_type=dev, dtype=encode_dtype):
                latents = model.get_reconstruction_latents(images)
        recon = model.get_latents_decoded_images(latents)
        recon_denorm = torch.clamp(denormalize(recon.float()), 0, 1)
        orig_denorm = torch.clamp(denormalize(images.float()), 0, 1)
    psnr = [calculate_psnr(orig_denorm[i] * 255.0, recon_denorm[i] * 255.0) for i in range(images.size(0))]
    return latents, recon_denorm, psnr


# ------------------------------------------------------------------------------------------------ linear probing (tools/test_linear_probing_hf.py)
class FeatureExtractor(nn.Module):
    """:109-131 -- get_intermediate_layers_feature(images, n = n_last_blocks, return_class_token = True) under inference_mode (and an
    autocast context the restatement leaves to the caller: the reference pins device_type='cuda')."""

    def __init__(self, model, n_last_blocks: int):
        super().__init__()
        self.model, self.n_last_blocks = model, n_last_blocks

    def forward(self, images):
        with torch.inference_mode():
            return self.model.get_intermediate_layers_feature(images, n=self.n_last_blocks, return_class_token=True)


def create_linear_input(x_tokens_list, use_n_blocks: int, use_avgpool: bool) -> torch.Tensor:
    """:137-152"""
    intermediate_output = x_tokens_list[-use_n_blocks:]
    output = torch.cat([class_token for _, class_token in intermediate_output], dim=-1)
    if use_avgpool:
        output = torch.cat((output, torch.mean(intermediate_output[-1][0], dim=1)), dim=-1)
        output = output.reshape(output.shape[0], -1)
    return output.float()


class LinearClassifier(nn.Module):
    """:155-170"""

    def __init__(self, out_dim: int, use_n_blocks: int, use_avgpool: bool, num_classes: int = 1000):
        super().__init__()
        self.use_n_blocks, self.use_avgpool = use_n_blocks, use_avgpool
        self.linear = nn.Linear(out_dim, num_classes)
        self.linear.weight.data.normal_(mean=0.0, std=0.01)
        self.linear.bias.data.zero_()

    def forward(self, x_tokens_list):
        return self.linear(create_linear_input(x_tokens_list, self.use_n_blocks, self.use_avgpool))


def probe_train_steps(feature_model, classifiers: Dict[str, nn.Module], batches, lr: float = 0.1) -> List[float]:
    """train_one_epoch (:257-299) without the progress bar / scheduler: per batch features = feature_model(images), outputs of every
    classifier, loss = sum of cross-entropies, SGD(momentum 0.9, weight_decay 0) step (:487-489; the cosine schedule is left out)."""
    params = [p for c in classifiers.values() for p in c.parameters()]
    opt = torch.optim.SGD(params, lr=lr, momentum=0.9, weight_decay=0)
    crit = nn.CrossEntropyLoss()
    losses = []
    for images, labels in batches:
        features = feature_model(images)  # inference tensors; create_linear_input's cat (outside inference mode) yields normal ones
        loss = sum(crit(c(features), labels) for c in classifiers.values())
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    return losses


# ------------------------------------------------------------------------------------------------ the oracle behind the reference's method surface
class OracleModel:
    """oracle/vtp_oracle.py's functions behind the method names the tools call (fp32 on CPU; with autocast_dtype every MODEL call runs
    under torch.autocast -- the tools' own arithmetic (classifier training, metrics) stays outside, as in the reference tools where the
    autocast context wraps the model call only) -- what the restated plumbing runs on where neither the reference tree nor a GPU is
    available, and the source of E_ref in the GPU test."""

    def __init__(self, sd: Dict[str, torch.Tensor], vis_heads: int, dec_heads: int, txt_heads: int, autocast_dtype=None):
        self.sd, self.hv, self.hd, self.ht, self.ac = sd, vis_heads, dec_heads, txt_heads, autocast_dtype

    def _run(self, fn):
        if self.ac is None:
            return fn()
        with torch.autocast("cpu", dtype=self.ac):
            out = fn()
        f32 = lambda t: t.float() if torch.is_tensor(t) else type(t)(f32(u) for u in t)
        return f32(out)

    def get_clip_text_feature(self, text, normalize=True):
        return self._run(lambda: O.clip_text_feature(self.sd, text, self.ht, normalize))

    def get_clip_image_feature(self, image, normalize=True):
        return self._run(lambda: O.clip_image_feature(self.sd, image, self.hv, normalize))

    def get_reconstruction_latents(self, image):
        return self._run(lambda: O.reconstruction_latents(self.sd, image, self.hv))

    def get_latents_decoded_images(self, latents):
        return self._run(lambda: O.decoder_forward(self.sd, latents.float(), self.hd))

    def get_intermediate_layers_feature(self, image, n=1, reshape=False, return_class_token=False, norm=True):
        return self._run(lambda: O.intermediate_layers(self.sd, image, self.hv, n=n, reshape=reshape,
                                                       return_class_token=return_class_token, norm=norm))


def run_all(model, device, images: torch.Tensor, targets: torch.Tensor, vocab: int, ctx: int, seed: int = 0) -> Dict[str, torch.Tensor]:
    """every restated tool path on one batch -> flat dict of tensors (the golden fixture layout)"""
    out = {}
    tok = toy_tokenizer(vocab, ctx)
    clf = build_zero_shot_classifier(model, tok, CLASSNAMES, TEMPLATES, num_classes_per_batch=3, device=device)
    t1, t5, logits = zero_shot_evaluate(model, clf, [(images[:4], targets[:4]), (images[4:], targets[4:])], device)
    out["zs.classifier"], out["zs.logits"], out["zs.top"] = clf.float().cpu(), logits, torch.tensor([t1, t5])
    lat, rec, psnr = reconstruct_and_psnr(model, images.to(device))
    out["rec.latents"], out["rec.recon_denorm"], out["rec.psnr"] = lat.float().cpu(), rec.float().cpu(), torch.tensor(psnr)
    fe = FeatureExtractor(model, n_last_blocks=2)
    feats = fe(images.to(device))
    for i, (p, c) in enumerate(feats):
        out[f"lp.patch{i}"], out[f"lp.cls{i}"] = p.float().cpu(), c.float().cpu()
    out["lp.input_1_avg"] = create_linear_input(feats, 1, True).cpu()
    out["lp.input_2"] = create_linear_input(feats, 2, False).cpu()
    torch.manual_seed(seed)
    D = out["lp.cls0"].shape[-1]
    clfs = {"blocks_1_avgpool_True": LinearClassifier(2 * D, 1, True, len(CLASSNAMES)).to(device),
            "blocks_2_avgpool_False": LinearClassifier(2 * D, 2, False, len(CLASSNAMES)).to(device)}
    losses = probe_train_steps(fe, clfs, [(images[:4].to(device), targets[:4].to(device)), (images[4:].to(device), targets[4:].to(device))] * 2)
    out["lp.losses"] = torch.tensor(losses)
    out["lp.w_after"] = clfs["blocks_1_avgpool_True"].linear.weight.detach().float().cpu()
    return out
