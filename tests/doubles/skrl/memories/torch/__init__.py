import torch


class Memory:
    def __init__(self, memory_size, num_envs=1, device=None, export=False, export_format="pt", export_directory=""):
        self.memory_size, self.num_envs = int(memory_size), int(num_envs)
        self.device = torch.device(device if device is not None else "cpu")
        self.tensors = {}
        self.memory_index, self.filled = 0, False

    def __len__(self):
        return (self.memory_size if self.filled else self.memory_index) * self.num_envs

    def create_tensor(self, name, size, dtype=torch.float32, keep_dimensions=False):
        size = int(size if isinstance(size, int) else torch.tensor(getattr(size, "shape", size)).prod())
        self.tensors[name] = torch.zeros(self.memory_size, self.num_envs, size, device=self.device, dtype=dtype)
        return True

    def add_samples(self, **tensors):
        for name, t in tensors.items():
            if name in self.tensors and t is not None:
                self.tensors[name][self.memory_index].copy_(t.reshape(self.num_envs, -1))
        self.memory_index += 1
        if self.memory_index >= self.memory_size:
            self.memory_index, self.filled = 0, True

    def get_tensor_by_name(self, name, keepdim=True):
        return self.tensors[name] if keepdim else self.tensors[name].view(-1, self.tensors[name].shape[-1])

    def set_tensor_by_name(self, name, tensor):
        self.tensors[name].copy_(tensor.reshape(self.tensors[name].shape))

    def sample_all(self, names, mini_batches=1, sequence_length=1):
        n = self.memory_size * self.num_envs
        idx = torch.randperm(n, device=self.device)
        for chunk in torch.chunk(idx, mini_batches):
            yield [self.tensors[k].view(n, -1)[chunk] for k in names]


class RandomMemory(Memory):
    def __init__(self, memory_size, num_envs=1, device=None, export=False, export_format="pt", export_directory="",
                 replacement=True):
        super().__init__(memory_size, num_envs, device, export, export_format, export_directory)
