"""Small host-side helpers with the reference's names (util.py:6-111): checkpoint I/O, JSON dump of the
run parameters, the step learning-rate schedule and the per-class loss weights."""
import json
import os
import shutil
import sys


def mkdir_if_not_exist(dirname):
    os.makedirs(dirname, exist_ok=True)


def yes_no_input():
    while True:
        choice = input("Please respond with 'yes' or 'no' [y/N]: ").lower()
        if choice in ("y", "ye", "yes"):
            return True
        if choice in ("n", "no", ""):
            return False


def check_if_done(filename):
    """The reference asks on stdin before overwriting an existing run (util.py:21-25).  Batch jobs have no
    terminal: when stdin is not a TTY the run proceeds and says so."""
    if os.path.exists(filename):
        print("%s already exists. Is it O.K. to overwrite it and start this program?" % filename)
        if not sys.stdin.isatty():
            print("(stdin is not a terminal: overwriting)")
            return
        if not yes_no_input():
            raise Exception("Please restart training after you set args.savename differently!")


def save_checkpoint(state, is_best, filename="checkpoint.pth.tar"):
    import torch
    torch.save(state, filename)
    if is_best:
        shutil.copyfile(filename, "model_best.pth.tar")


def load_checkpoint(filename, map_location="cpu"):
    """torch.load of a reference-layout checkpoint (the dict pickles an argparse.Namespace under 'args')."""
    import torch
    return torch.load(filename, map_location=map_location, weights_only=False)


def calc_entropy(output):
    import torch
    import torch.nn.functional as F
    output = F.softmax(output, dim=1)
    return -torch.mean(output * torch.log(output + 1e-6))


class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def save_dic_to_json(dic, fn, verbose=True):
    dic = {str(k): v for k, v in dic.items()}
    text = json.dumps(dic, sort_keys=True, indent=4, default=str)
    if verbose:
        print(text)
    with open(fn, "w") as f:
        f.write(text)
    print("param file '%s' was saved!" % fn)


def emphasize_str(string):
    print("#" * 100)
    print(string)
    print("#" * 100)


def adjust_learning_rate(optimizer, lr_init, decay_rate, epoch, num_epochs):
    """lr * decay at 1/2 of the epochs, lr * decay^2 at 3/4 (util.py:87-96).  The trainers pass
    ``args.weight_decay`` as ``decay_rate`` (adapt_trainer.py:228-230) -- reproduced by the callers."""
    lr = lr_init
    if epoch >= num_epochs * 0.75:
        lr *= decay_rate ** 2
    elif epoch >= num_epochs * 0.5:
        lr *= decay_rate
    for group in optimizer.param_groups:
        group["lr"] = lr
    return lr


def get_class_weight_from_file(n_class, weight_filename=None, add_bg_loss=False):
    """ones(n_class), optionally scaled by a CSV with columns class_id, weight; the last class (background)
    is zeroed unless ``add_bg_loss`` (util.py:99-111)."""
    import torch
    weight = torch.ones(n_class)
    if weight_filename:
        import pandas as pd
        df = pd.read_csv(weight_filename).sort_values("class_id")
        weight *= torch.tensor(df.weight.values, dtype=torch.float32)
    if not add_bg_loss:
        weight[n_class - 1] = 0
    return weight
