"""Checkpoint files in the reference's format (``ssds/core/checkpoint.py``): a plain ``state_dict``
saved as ``<prefix>_epoch_<n>.pth`` plus a ``checkpoint_list.txt`` index (save :18-35, find :38-56), and a
tolerant loader that strips ``module.``, filters by scope and loads whatever matches (resume :59-133), so
weights trained with the reference load here and vice versa."""
import os
from collections import OrderedDict

import torch


def save_checkpoints(model, output_dir, checkpoint_prefix, epochs):
    if not os.path.exists(output_dir):
        os.makedirs(output_dir)
    filename = checkpoint_prefix + "_epoch_{:d}".format(epochs) + ".pth"
    filename = os.path.join(output_dir, filename)
    state = model.module.state_dict() if hasattr(model, "module") else model.state_dict()
    torch.save(OrderedDict((k, v.detach().cpu()) for k, v in state.items()), filename)
    with open(os.path.join(output_dir, "checkpoint_list.txt"), "a") as f:
        f.write("epoch {epoch:d}: {filename}\n".format(epoch=epochs, filename=filename))
    print("Wrote snapshot to: {:s}".format(filename))
    return filename


def find_previous_checkpoint(output_dir):
    """-> ([epochs], [paths]) parsed from checkpoint_list.txt, or False when there is none."""
    path = os.path.join(output_dir, "checkpoint_list.txt")
    if not os.path.exists(path):
        return False
    epoches, resume_checkpoints = [], []
    with open(path, "r") as f:
        for line in f.read().splitlines():
            epoch = int(line[line.find("epoch ") + len("epoch "): line.find(":")])
            checkpoint = line[line.find(":") + 2:]
            epoches.append(epoch)
            resume_checkpoints.append(checkpoint)
    return epoches, resume_checkpoints


def resume_checkpoint(model, resume_checkpoint, resume_scope=""):
    """Partial, scope-filtered load; returns the model.  Keys that do not exist in the model (e.g. the
    reference backbone's unused ``head_conv`` / ``classifier``) or whose shape differs are skipped and
    reported."""
    if resume_checkpoint == "" or not os.path.isfile(resume_checkpoint):
        print(("=> no checkpoint found at '{}'".format(resume_checkpoint)))
        return False
    print(("=> loading checkpoint '{:s}'".format(resume_checkpoint)))
    checkpoint = torch.load(resume_checkpoint, map_location=torch.device("cpu"))
    if "state_dict" in checkpoint:
        checkpoint = checkpoint["state_dict"]
    checkpoint = OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in checkpoint.items())
    if resume_scope != "":
        scopes = resume_scope.split(",")
        checkpoint = OrderedDict((k, v) for k, v in checkpoint.items() if any(s in k for s in scopes))
    target = model.module if hasattr(model, "module") else model
    own = target.state_dict()
    loaded, skipped = OrderedDict(), []
    for k, v in checkpoint.items():
        if k in own and tuple(own[k].shape) == tuple(v.shape):
            loaded[k] = v
        else:
            skipped.append(k)
    missing = [k for k in own if k not in loaded]
    if skipped:
        print("=> skipped (absent or shape mismatch): {}".format(skipped[:8] + (["..."] if len(skipped) > 8 else [])))
    if missing:
        print("=> not in checkpoint: {} keys".format(len(missing)))
    own.update(loaded)
    target.load_state_dict(own)
    return model
