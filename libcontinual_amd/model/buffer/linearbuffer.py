"""Plain rehearsal containers (reference core/model/buffer/linearbuffer.py:4-27); the trainer fills them
according to `strategy` (core/trainer.py:410-418).  Attribute names are the reference's (the trainer's rehearsal merge and
the update functions read them directly)."""

__all__ = ["LinearBuffer", "LinearSpiltBuffer"]


class _Rehearsal:
    """capacity bookkeeping shared by the containers: `buffer_size` exemplars in total, `total_classes` seen so far"""

    def __init__(self, capacity, policy, batch):
        self.buffer_size, self.strategy, self.batch_size = capacity, policy, batch
        self.total_classes = 0

    def per_class_quota(self):
        return self.buffer_size // max(1, self.total_classes)


class LinearBuffer(_Rehearsal):
    """`images` (paths or array indices) and `labels`, parallel lists"""

    def __init__(self, buffer_size, strategy, batch_size):
        super().__init__(buffer_size, strategy, batch_size)
        self.images, self.labels = [], []

    def __len__(self):
        return len(self.labels)

    def is_empty(self):
        return not self.labels


class LinearSpiltBuffer(_Rehearsal):
    """train / validation split variant; like the reference it fixes the ratio at 0.1 whatever `val_ratio` says"""

    def __init__(self, buffer_size, strategy, batch_size, val_ratio):
        super().__init__(buffer_size, strategy, batch_size)
        self.val_ratio = 0.1
        self.train_images, self.train_labels, self.val_images, self.val_labels = [], [], [], []

    def is_empty(self):
        return not self.train_labels
