"""HostPool: the per-molecule host stage of the step (validity, reward) spread over processes keeps
order and equals the serial loop of the reference (reinforce.py:195-196), failing items included."""
import numpy as np


def _valid(s):
    if s == 'boom':
        raise ValueError('parser error')
    return len(s) % 3 != 0


def _score(s):
    return min(1.0, s.count('C') / 4.0)


def test_pool_equals_the_serial_loop_and_keeps_order():
    from paccmann_generator_amd.hostpool import HostPool
    rng = np.random.RandomState(0)
    smiles = [''.join(rng.choice(list('CNO=()'), size=rng.randint(0, 30))) for _ in range(1000)]
    smiles[17] = 'boom'
    serial = []
    for s in smiles:
        try:
            serial.append(_valid(s))
        except ValueError:
            serial.append(False)
    with HostPool(_valid, workers=3, chunk=64, on_error=False) as pool:
        assert pool(smiles) == serial
        assert pool([]) == []
        assert pool(smiles[:5]) == serial[:5]              # a second call on the same workers
    with HostPool(_score, workers=2, chunk=100, on_error=0.0) as pool:
        assert pool(smiles) == [0.0 if s == 'boom' else _score(s) for s in smiles]
    one = HostPool(_valid, workers=1, on_error=False)          # no processes at all
    assert one(smiles) == serial
    one.close()


def test_pool_plugs_into_the_validity_hook_signature():
    """is_valid_batch(list[str]) -> list[bool] is the hook Reinforce.get_smiles_from_latent calls."""
    from paccmann_generator_amd.hostpool import HostPool
    with HostPool(_valid, workers=2, chunk=2, on_error=False) as pool:
        out = pool(['CC', 'CCC', 'boom', 'C'])
    assert out == [True, False, False, True] and all(isinstance(v, bool) for v in out)
