"""GPT training data path: `.vq.pth` code files + text token ids -> the collater dict the trainer consumes.

Mirrors ttts/gpt/dataset.py: `GptTtsDataset.__getitem__` (:36-59: returns `(text, qmel, wav_length)` or None for bad
items, drops text > 400 / codes > 600) and `GptTtsCollater.__call__` (:65-97: filters None, right-pads text and codes
with 0, returns `padded_text, text_lengths, padded_qmel, qmel_lengths, wav_lens`).  Out of scope and therefore injected:
the pinyin + BPE tokenisation (pypinyin / a trained tokenizer) and `torchaudio.load`; items carry ready token ids and
the wav length in samples at 24 kHz.
"""
import json

import torch
import torch.nn.functional as F
from torch import LongTensor


class GptTtsDataset(torch.utils.data.Dataset):
    """jsonl lines: {"path": <audio path>, "text_ids": [...], "wav_length": <samples at 24 kHz>}; codes in `<path>.vq.pth`."""

    def __init__(self, jsonl_path, max_text=400, max_codes=600):
        with open(jsonl_path, encoding="utf8") as f:
            self.items = [json.loads(line) for line in f if line.strip()]
        self.max_text, self.max_codes = max_text, max_codes

    def __getitem__(self, index):
        try:
            it = self.items[index]
            text = LongTensor(it["text_ids"])
            qmel = LongTensor(torch.load(it["path"] + ".vq.pth"))
            wav_length = int(it["wav_length"])
        except Exception as e:                       # the reference prints and returns None (dataset.py:49-51)
            print(e)
            return None
        if text.shape[0] > self.max_text or qmel.shape[0] > self.max_codes:
            return None
        return text, qmel, wav_length

    def __len__(self):
        return len(self.items)


class GptTtsCollater:
    def __init__(self, cfg=None):
        self.cfg = cfg

    def __call__(self, batch):
        batch = [x for x in batch if x is not None]
        if len(batch) == 0:
            return None
        text_lens = [len(x[0]) for x in batch]
        qmel_lens = [len(x[1]) for x in batch]
        wav_lens = [x[2] for x in batch]
        max_text_len, max_qmel_len = max(text_lens), max(qmel_lens)
        texts = [F.pad(t, (0, max_text_len - len(t)), value=0) for t, _, _ in batch]
        qmels = [F.pad(q, (0, max_qmel_len - len(q)), value=0) for _, q, _ in batch]
        return {"padded_text": torch.stack(texts), "text_lengths": LongTensor(text_lens), "padded_qmel": torch.stack(qmels),
                "qmel_lengths": LongTensor(qmel_lens), "wav_lens": LongTensor(wav_lens)}
