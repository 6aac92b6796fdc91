"""CPU oracle for the REINFORCE hot path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the decoder arithmetic of the reference lives in the third-party
package ``paccmann_chemistry`` (pins: @0.0.4 in /root/reference/requirements.txt:3,
@0.0.1 in examples/IC50/requirements.txt:3, @sarscov2 in
examples/affinity/requirements.txt:3), which is neither vendored under
/root/reference nor installed here, and the reference holds no test, golden vector
or fixture for this path.  The oracle therefore restates the
published algorithm of that package from ``torch.nn`` primitives and anchors on the
reference's own call sites (paccmann_generator/reinforce.py:149-211,
reinforce_omics.py:276-342, reinforce_proteins.py:267-331).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  The product path
(``paccmann_generator_amd``) never does.
"""
