"""CPU: the data layer (itr_amd/datamodule) against fixtures captured from the reference's own
itr/datamodule/{data_loader,vocab,tokenization}.py (tests/golden/g14_data_layer.npz, oracle/make_goldens.py g14).
Integer work: everything is compared exactly.  nltk is absent in this image, so the fixture was made with the
regex word tokeniser named in its `tokenizer_note`; the same function is the product's fallback."""
import json
import os

import numpy as np
import pytest
import torch

from itr_amd.datamodule import data_loader as dl, tokenization as tok, vocab as vocab_mod


def _text(arr):
    return bytes(np.asarray(arr, np.uint8)).decode('utf-8')


@pytest.fixture()
def toy(golden, tmp_path):
    g = golden("g14_data_layer")
    name = 'toy_precomp'
    d = tmp_path / name
    d.mkdir()
    for split in ('train', 'dev', 'test'):
        np.save(d / ('%s_ims.npy' % split), g["ims"])
        np.save(d / ('%s_boxes.npy' % split), g["boxes"])
        np.save(d / ('%s_img_sizes.npy' % split), g["img_sizes"])
        (d / ('%s_caps.txt' % split)).write_bytes(bytes(g["caps_blob"]))
    vdir = tmp_path / 'vocab'
    vdir.mkdir()
    (vdir / ('%s_vocab.json' % name)).write_text(_text(g["vocab_json"]))
    vfile = tmp_path / 'bert_vocab.txt'
    vfile.write_bytes(bytes(g["bert_vocab"]))
    return g, str(tmp_path), str(d), str(vdir), str(vfile), name


def test_build_vocab_matches_reference(toy):
    g, root, d, vdir, vfile, name = toy
    v = vocab_mod.build_vocab(root, name, caption_file={name: ['train_caps.txt']}, threshold=2,
                              tokenize=tok.regex_word_tokenize)
    ref = json.loads(_text(g["vocab_json"]))
    assert v.word2idx == ref['word2idx'] and v.idx == ref['idx'] and len(v) == int(g["vocab_len"])
    assert [v('<pad>'), v('<start>'), v('<end>'), v('<unk>')] == [0, 1, 2, 3]
    assert v('never-seen-word') == 3
    out = os.path.join(root, 'round_trip.json')
    vocab_mod.serialize_vocab(v, out)
    v2 = vocab_mod.deserialize_vocab(out)
    assert v2.word2idx == v.word2idx and v2('zebra') == v('zebra')


def _cfg(vdir, name, **kw):
    cfg = {'use_bbox': False, 'text_encoder': 'gru', 'vocab_path': vdir, 'data_name': name, 'vocab_type': 'json',
           'name': 'SCAN', 'word_tokenize': tok.regex_word_tokenize}
    cfg.update(kw)
    return cfg


def test_precomp_dataset_items_and_collate(toy):
    g, root, d, vdir, vfile, name = toy
    ds = dl.PrecompDataset(d, 'test', _cfg(vdir, name, ref_quirk_bytes_repr=True))
    assert len(ds) == int(g["test_len"]) and ds.im_div == int(g["test_im_div"]) == 5
    assert len(dl.PrecompDataset(d, 'dev', _cfg(vdir, name))) == int(g["dev_len"]) == 5000   # data_loader.py:78-80
    lens = g["gru_ids_len"]
    want = np.split(g["gru_ids_concat"], np.cumsum(lens)[:-1])
    for i in range(len(ds)):
        assert ds[i][3].tolist() == want[i].tolist(), i
    it = ds[7]
    assert np.array_equal(it[0].numpy(), g["item7_image"]) and [it[4], it[5]] == g["item7_meta"].tolist()
    assert it[1] is None and it[2] is None and it[6] is None and it[7] is None
    batch = dl.collate_fn([ds[int(i)] for i in g["pick"]])
    assert np.array_equal(batch[0].numpy(), g["col_images"])
    assert np.array_equal(batch[3].numpy(), g["col_ids"]) and batch[3].dtype == torch.long
    assert list(batch[4]) == g["col_lengths"].tolist() and list(batch[5]) == g["col_index"].tolist()
    assert sorted(batch[4], reverse=True) == list(batch[4])
    assert set(batch[1]) == {None} and set(batch[6]) == {None}


def test_utf8_decode_is_the_default(toy):
    """Without the compat switch captions are decoded, so the first word is itself and not  b'<word>."""
    g, root, d, vdir, vfile, name = toy
    ds = dl.PrecompDataset(d, 'test', _cfg(vdir, name))
    ids = ds.token_ids(0)          # "A man riding a wave on top of a surfboard ."
    assert ids[0] == ds.vocab('<start>') and ids[-1] == ds.vocab('<end>')
    assert ids[1] == ds.vocab('a') and ids[2] == ds.vocab('man') and len(ids) == 11 + 2
    quirk = dl.PrecompDataset(d, 'test', _cfg(vdir, name, ref_quirk_bytes_repr=True)).token_ids(0)
    assert len(quirk) == len(ids) + 3      # b ' ... '


def test_wordpiece_tokenizer_and_features(toy):
    g, root, d, vdir, vfile, name = toy
    tk = tok.FullTokenizer(vocab_file=vfile, do_lower_case=True)
    sentences = _text(g["bert_sentences"]).split("\n")
    want_tokens = _text(g["bert_tokens"]).split("\n")
    assert len(sentences) == len(want_tokens) == len(g["bert_input_ids"])
    for sn, wt, ids, mask, types in zip(sentences, want_tokens, g["bert_input_ids"], g["bert_input_mask"], g["bert_type_ids"]):
        assert " ".join(tk.tokenize(sn)) == wt, sn
        _, got_ids, got_mask, got_types = dl.convert_to_feature(sn.encode('utf-8'), 12, tk)
        assert got_ids == ids.tolist() and got_mask == mask.tolist() and got_types == types.tolist(), sn
    assert tk.tokenize("unaffable") == ["un", "##aff", "##able"]


def test_bert_dataset_with_boxes(toy):
    g, root, d, vdir, vfile, name = toy
    cfg = {'use_bbox': True, 'text_encoder': 'bert', 'max_words': 12, 'vocab_file': vfile, 'data_name': name, 'name': 'CAMERA'}
    ds = dl.PrecompDataset(d, 'test', cfg)
    b = dl.collate_fn([ds[int(i)] for i in g["pick"]])
    for got, key in zip((b[0], b[1], b[2], b[3], b[6], b[7]), ("bcol_images", "bcol_boxes", "bcol_wh", "bcol_ids", "bcol_mask", "bcol_types")):
        assert np.array_equal(got.numpy(), g[key]), key
    assert [int(x) for x in b[4]] == g["bcol_lengths"].tolist() and b[5].tolist() == g["bcol_index"].tolist()


def test_loader_contract(toy):
    g, root, d, vdir, vfile, name = toy
    cfg = _cfg(vdir, name, data_path=root)
    loader, vocab_size = dl.get_test_loader('test', name, 8, 0, cfg)
    assert vocab_size == int(g["vocab_len"])
    seen = []
    for images, boxes, imgs_wh, captions_ids, lengths, ids, mask, types in loader:
        assert images.shape[1:] == (36, 8) and captions_ids.shape == (len(ids), max(lengths))
        seen.extend(ids)
    assert sorted(seen) == list(range(30))
    with pytest.raises(NotImplementedError):
        dl.get_loaders('coco', 8, 0, dict(cfg, data_name='coco'))


def test_vsrn_caption_layout_matches_reference(toy):
    """data_loader.py:117-125 as written: every caption max_len + 1 ids (truncation drops <end>, <pad> tail), a mask
    computed after the padding, collate_fn lengths all max_len + 1."""
    g, root, d, vdir, vfile, name = toy
    ds = dl.PrecompDataset(d, 'test', dict(_cfg(vdir, name, max_len=9, ref_quirk_bytes_repr=True), name='VSRN'))
    ids = np.stack([ds[i][3].numpy() for i in range(len(ds))])
    mask = np.stack([ds[i][6].numpy() for i in range(len(ds))])
    assert ids.dtype == np.int64 and (ids == g["vsrn_ids"]).all() and (mask == g["vsrn_mask"]).all()
    pick = [int(i) for i in g["pick"]]
    b = dl.collate_fn([ds[i] for i in pick])
    assert (b[3].numpy() == g["vcol_ids"]).all() and list(b[4]) == list(g["vcol_lengths"]) and list(b[5]) == list(g["vcol_index"])
    assert (b[6].numpy() == g["vcol_mask"]).all() and b[7] == (None,) * len(pick)
