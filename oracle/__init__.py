"""CPU oracle for the SSL4POLYP MAE / ViT-B/16 hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``ssl4polyp_amd/`` (the product) may
import from this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and only as the checker / the timed
CPU baseline -- never as the thing measured or shipped.

What it is: a plain fp32 PyTorch-CPU + numpy restatement of the reference's
algorithm for this path, each function citing the reference file:line it
follows (paths relative to the reference checkout).

Third-party arithmetic the reference delegates to and that is NOT under the
reference tree: ``timm==0.4.12`` (requirements.txt:5) -- ``PatchEmbed``,
``Attention``, ``Mlp``, ``Block``, ``VisionTransformer``.  Its published
algorithm is restated in ``vit_mae_ref.py``.

Parity pinning: the reference has no golden vectors for this path (its tests
never import the model files).  The oracle is pinned by fixtures generated in
the build container by importing the reference's own ``models_mae.py`` /
``models.py`` / ``pos_embed.py`` / ``lr_sched.py`` (with an in-process stand-in
for the two absent timm classes) -- see ``tests/golden/make_fixtures.py`` -- and
cross-checked against the independent ViT-MAE implementation in the installed
``transformers`` package.
"""
