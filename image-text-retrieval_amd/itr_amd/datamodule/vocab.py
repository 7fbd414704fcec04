"""Vocabulary wrapper with the reference's JSON layout (itr/datamodule/vocab.py:28-68):
{"word2idx": {...}, "idx2word": {"0": "<pad>", ...}, "idx": N}; ids 0..3 are <pad>, <start>, <end>, <unk>
(vocab.py:98-101).  Unknown words map to <unk> (vocab.py:43-46)."""
import json
import os
from collections import Counter

from . import tokenization

annotations = {   # caption files per dataset (vocab.py:17-25)
    'coco_precomp': ['train_caps.txt', 'dev_caps.txt'],
    'coco': ['annotations/captions_train2014.json', 'annotations/captions_val2014.json'],
    'f8k_precomp': ['train_caps.txt', 'dev_caps.txt'],
    '10crop_precomp': ['train_caps.txt', 'dev_caps.txt'],
    'f30k_precomp': ['train_caps.txt', 'dev_caps.txt'],
    'f8k': ['dataset_flickr8k.json'],
    'f30k': ['dataset_flickr30k.json'],
}


class Vocabulary(object):
    def __init__(self):
        self.word2idx = {}
        self.idx2word = {}
        self.idx = 0

    def add_word(self, word):
        if word not in self.word2idx:
            self.word2idx[word] = self.idx
            self.idx2word[self.idx] = word
            self.idx += 1

    def __call__(self, word):
        return self.word2idx.get(word, self.word2idx['<unk>'])

    def __len__(self):
        return len(self.word2idx)


def serialize_vocab(vocab, dest):
    with open(dest, "w") as f:
        json.dump({'word2idx': vocab.word2idx, 'idx2word': vocab.idx2word, 'idx': vocab.idx}, f)


def deserialize_vocab(src):
    with open(src) as f:
        d = json.load(f)
    vocab = Vocabulary()
    vocab.word2idx, vocab.idx2word, vocab.idx = d['word2idx'], d['idx2word'], d['idx']
    return vocab


def from_txt(txt):
    with open(txt, 'rb') as f:
        return [line.strip() for line in f]


def build_vocab(data_path, data_name, caption_file=annotations, threshold=4, tokenize=None):
    """Words seen at least `threshold` times, after the 4 special tokens, in first-seen order (vocab.py:78-106)."""
    tokenize = tokenize or tokenization.word_tokenize
    counter = Counter()
    for path in caption_file[data_name]:
        for caption in from_txt(os.path.join(data_path, data_name, path)):
            counter.update(tokenize(caption.lower().decode('utf-8')))
    vocab = Vocabulary()
    for special in ('<pad>', '<start>', '<end>', '<unk>'):
        vocab.add_word(special)
    for word, cnt in counter.items():
        if cnt >= threshold:
            vocab.add_word(word)
    return vocab
