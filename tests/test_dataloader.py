"""CPU: native data ingest + drop-in loaders against what the reference's own loaders
expose on the committed fixture data set tests/golden/ds_small (goldens g8, tests/golden/gen_goldens.py)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from invpref_kdd_2022_amd import dataloader as dl

G = os.path.join(os.path.dirname(__file__), 'golden')
CPU = torch.device('cpu')


def _csr_of(sets):
    ptr, idx = np.zeros(len(sets) + 1, np.int64), []
    for i, s in enumerate(sets):
        idx.extend(sorted(s))
        ptr[i + 1] = len(idx)
    return ptr, np.array(idx, np.int64)


def test_ingest_library_exports_the_header():
    import re
    hdr = open(os.path.join(os.path.dirname(G), '..', 'include', 'invpref_ingest.h')).read()
    names = set(re.findall(r'\b(invpref_\w+)\s*\(', hdr))
    assert names == {'invpref_ingest_abi_version', 'invpref_csv_shape', 'invpref_csv_read_f64', 'invpref_csr_sets'}
    L = dl.lib()
    for n in names:
        getattr(L, n)
    assert L.invpref_ingest_abi_version() == 1


def test_implicit_loader_matches_reference():
    z = np.load(os.path.join(G, 'g8_loader_implicit.npz'))
    ld = dl.YahooUniformImplicitBCELossDataLoader(os.path.join(G, 'ds_small', 'implicit'), CPU, has_item_pool_file=True)
    np.testing.assert_array_equal(ld.train_data_np, z['train_data_np'])
    assert ld.train_data_np.dtype == np.int64
    np.testing.assert_array_equal(ld.test_data_np, z['test_data_np'])
    np.testing.assert_array_equal(ld.uniform_data_np, z['uniform_data_np'])
    assert [ld.user_num, ld.item_num, ld.train_data_len, ld.test_data_len, ld.uniform_data_len] == list(z['dims'])
    assert ld.all_test_users_by_sorted_list == list(z['test_user_list'])
    np.testing.assert_array_equal(ld.all_test_users_by_sorted_tensor.numpy(), z['test_users_tensor'])
    assert ld.user_list == list(z['user_list']) and ld.item_list == list(z['item_list'])
    assert ld.test_item_list == list(z['test_item_list'])
    for attr, key in (('user_positive_interaction', 'pos'), ('ground_truth', 'truth'), ('item_pool', 'pool')):
        sets = getattr(ld, attr)
        np.testing.assert_array_equal(sets.indptr, z[key + '_ptr'])
        np.testing.assert_array_equal(sets.indices, z[key + '_idx'])
        ptr, idx = _csr_of(list(sets))          # the list-of-sets view says the same
        np.testing.assert_array_equal(ptr, z[key + '_ptr'])
        np.testing.assert_array_equal(idx, z[key + '_idx'])
    ptr, idx = _csr_of(ld.get_sorted_all_test_users_ground_truth)
    np.testing.assert_array_equal(ptr, z['sorted_truth_ptr'])
    np.testing.assert_array_equal(idx, z['sorted_truth_idx'])
    u0 = ld.all_test_users_by_sorted_list[0]
    assert isinstance(ld.user_mask_items(u0), set) and isinstance(ld.user_highlight_items(u0), set)
    assert ld.get_user_ground_truth(u0) == set(z['sorted_truth_idx'][z['sorted_truth_ptr'][0]:z['sorted_truth_ptr'][1]].tolist())
    with pytest.raises(IndexError):             # like the reference's list: users beyond the last positive one
        ld.user_positive_interaction[len(ld.user_positive_interaction)]
    # the CSR hand-over to the evaluator
    ev = ld.csr_for_eval()
    np.testing.assert_array_equal(ev['mask'][0], z['mask_of_test_users_ptr'])
    np.testing.assert_array_equal(ev['mask'][1], z['mask_of_test_users_idx'])
    np.testing.assert_array_equal(ev['truth'][0], z['sorted_truth_ptr'])
    np.testing.assert_array_equal(ev['truth'][1], z['sorted_truth_idx'])
    assert ev['mask'][0].dtype == np.int32 and ev['highlight'][1].dtype == np.int32
    no_pool = dl.YahooImplicitBCELossDataLoader(os.path.join(G, 'ds_small', 'implicit'), CPU)
    with pytest.raises(NotImplementedError):
        no_pool.user_highlight_items(u0)


def test_packed_round_trip(tmp_path):
    src = os.path.join(G, 'ds_small', 'implicit')
    a = dl.YahooImplicitBCELossDataLoader(src, CPU, has_item_pool_file=True)
    a.save_packed(str(tmp_path / 'packed'))
    assert np.load(tmp_path / 'packed' / 'train_ids.npy').dtype == np.int32
    assert np.load(tmp_path / 'packed' / 'train_scores.npy').dtype == np.int8
    b = dl.YahooImplicitBCELossDataLoader.from_packed(str(tmp_path / 'packed'), CPU)
    np.testing.assert_array_equal(a.train_data_np, b.train_data_np)
    np.testing.assert_array_equal(a.test_data_np, b.test_data_np)
    assert (a.user_num, a.item_num) == (b.user_num, b.item_num)
    for attr in ('user_positive_interaction', 'ground_truth', 'item_pool'):
        np.testing.assert_array_equal(getattr(a, attr).indptr, getattr(b, attr).indptr)
        np.testing.assert_array_equal(getattr(a, attr).indices, getattr(b, attr).indices)


def test_explicit_loader_matches_reference():
    z = np.load(os.path.join(G, 'g8_loader_explicit.npz'))
    ld = dl.ExplicitUniformDataLoader(os.path.join(G, 'ds_small', 'explicit'), CPU)
    assert [ld.user_num, ld.item_num, ld.train_data_len, ld.test_data_len, ld.uniform_data_len] == list(z['dims'])
    for k in ('train_data_np', 'test_data_np', 'uniform_data_np', 'all_test_pairs_np', 'all_test_scores_np',
              'all_train_pairs_np', 'all_train_scores_np'):
        got = getattr(ld, k)
        np.testing.assert_array_equal(got, z[k])
        assert got.dtype == z[k].dtype, k
    for k in ('all_test_pairs_tensor', 'all_test_scores_tensor', 'test_data_tensor', 'all_train_pairs_tensor',
              'all_train_scores_tensor', 'train_data_tensor'):
        got = getattr(ld, k)
        np.testing.assert_array_equal(got.numpy(), z[k])
        assert got.numpy().dtype == z[k].dtype, k


def test_csv_edge_cases(tmp_path):
    def w(name, text):
        p = tmp_path / name
        p.write_bytes(text.encode())
        return str(p)
    # no header, CRLF, no trailing newline, blank lines, negative / fractional / exponent fields, spaces
    a = dl.read_csv(w('a.csv', '1,2,3\r\n\r\n-4, 5.25 ,6e2\n7,8,+9'), header=False)
    np.testing.assert_array_equal(a, [[1, 2, 3], [-4, 5.25, 600], [7, 8, 9]])
    np.testing.assert_array_equal(dl.read_csv(str(tmp_path / 'a.csv'), header='auto'), a)
    # the default mirrors pd.read_csv: the first line is the header even when it is numeric
    np.testing.assert_array_equal(dl.read_csv(str(tmp_path / 'a.csv')), a[1:])
    import pandas as pd
    np.testing.assert_array_equal(pd.read_csv(str(tmp_path / 'a.csv'), skipinitialspace=True).values, a[1:])
    assert dl.read_csv(w('one.csv', '1,2\n')).shape == (0, 2)
    with pytest.raises(dl.IngestError):
        dl.read_csv(w('hh.csv', 'u,i\n1,2\n'), header=False)                  # the header line is not data
    assert dl.read_csv(w('h.csv', 'user_id,item_id\n')).shape == (0, 2)       # header only
    assert dl.read_csv(w('e.csv', '')).shape == (0, 0)                         # empty file
    with pytest.raises(dl.IngestError):
        dl.read_csv(w('r.csv', 'u,i\n1,2\n3\n'))                              # ragged line
    with pytest.raises(dl.IngestError):
        dl.read_csv(w('x.csv', 'u,i\n1,abc\n'))                               # not a number
    with pytest.raises(dl.IngestError):
        dl.read_csv(str(tmp_path / 'missing.csv'))
    # many lines across several parser threads: identical to numpy's own parse
    rs = np.random.RandomState(5)
    big = rs.randint(0, 10 ** 6, (200000, 3))
    p = w('big.csv', 'a,b,c\n' + '\n'.join(','.join(map(str, r)) for r in big.tolist()) + '\n')
    np.testing.assert_array_equal(dl.read_csv(p), big.astype(np.float64))
    # exact decimal conversion (same doubles as python's float())
    vals = ['0.1', '3.14159', '123456.789012', '1e-5', '-2.5E+3', '0.000001', '99999999999999']
    p = w('f.csv', '\n'.join(vals) + '\n')
    np.testing.assert_array_equal(dl.read_csv(p, header=False)[:, 0], [float(v) for v in vals])
    # CSR: duplicates collapse, users without pairs get empty sets, bad ids are refused
    ptr, idx = dl.csr_sets(np.array([2, 0, 2, 2, 0]), np.array([5, 1, 5, 3, 0]), 4)
    assert ptr.tolist() == [0, 2, 2, 4, 4] and idx.tolist() == [0, 1, 3, 5]
    ptr, idx = dl.csr_sets(np.zeros(0, np.int64), np.zeros(0, np.int64), 3)
    assert ptr.tolist() == [0, 0, 0, 0] and len(idx) == 0
    with pytest.raises(dl.IngestError):
        dl.csr_sets(np.array([4]), np.array([1]), 4)
