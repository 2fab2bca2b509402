"""oracle/ -- CPU restatement of the reference's algorithms for the hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under nextbestpath_amd/ imports this package; only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and there only
as the checker / the reported CPU baseline -- never as the thing measured or shipped.

Pinning status (how each restatement is tied to the reference):
  * nbp_net.py, maps.py, planner.py, coverage.py: PINNED by golden vectors generated in the
    build container by importing the reference itself (tests/golden/make_golden.py; the
    reference has no tests / golden vectors of its own -- SURVEY.md section 4).
  * camera.py::ndc_tables, ::pose_lattice and the candidate mask / keep count of ::partial_point_cloud: PINNED (round 5) by
    tests/golden/camera.npz -- the reference's own Camera.__init__ tables (bit for bit), compute_partial_point_cloud with
    an identity un-projection, and obtain_depth's depth / mask / numpy draws (tests/test_oracle_golden.py).
  * the rest of camera.py (PyTorch3D camera conventions), raster.py (PyTorch3D rasteriser), mesh_rays.py
    (trimesh ray tests): PARITY UNPINNED -- pytorch3d 0.7.4 / trimesh 4.1.2 are third-party
    dependencies absent from /root/reference and from this image; they restate the
    libraries' documented conventions and are validated by analytic known-answer scenes.
"""
