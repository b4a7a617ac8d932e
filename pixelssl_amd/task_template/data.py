"""Task-dataset plugin ABI (pixelssl/task_template/data.py:14-98): `TaskDataset(args, is_train)` with `sample_list`,
`idxs`, `im_loader`, `root_dir` from args.trainset / args.valset, `__getitem__ -> (inputs tuple, labels tuple)`."""
from PIL import Image
from torch.utils.data import Dataset


def add_parser_arguments(parser):
    pass


def task_dataset():
    return TaskDataset


class ImageLoader:
    def load(self, name):
        return Image.open(name)


class TaskDataset(Dataset):
    def __init__(self, args=None, is_train=True):
        super().__init__()
        self.args = args
        self.is_train = is_train
        self.root_dir = None
        self.sample_list = []
        self.idxs = []
        self.im_loader = ImageLoader()
        sets = self.args.trainset if is_train else self.args.valset
        self.root_dir = list(sets.values())[0]

    def __len__(self):
        return len(self.sample_list)

    def __getitem__(self, idx):
        raise NotImplementedError
