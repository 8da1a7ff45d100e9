"""The C-ABI library loads on a machine without a GPU, exports every symbol that
include/szg/abi.h, include/szg/host.h, include/szg/raster.h and include/szg/assets.h declare, keeps the reference's struct sizes,
and fails loudly (no CPU fallback) when asked to create a pipeline without a device."""
import ctypes as C
import os
import re

import pytest

from syzygy_amd import abi, lib, library_path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header):
    text = open(os.path.join(ROOT, "include", "szg", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(szg_[a-z0-9_]+)\s*\(", text)))


def test_library_is_in_tree():
    assert os.path.exists(library_path())
    assert library_path().startswith(ROOT)


@pytest.mark.parametrize("header,table", [("abi.h", abi.ABI_FUNCTIONS), ("host.h", abi.HOST_FUNCTIONS),
                                          ("raster.h", abi.RASTER_FUNCTIONS), ("assets.h", abi.ASSET_FUNCTIONS)])
def test_every_declared_symbol_is_exported_and_bound(header, table):
    handle = C.CDLL(library_path())
    names = declared_symbols(header)
    assert names, header
    for name in names:
        assert hasattr(handle, name), f"{name} declared in {header} but not exported"
        assert name in table, f"{name} declared in {header} but has no ctypes signature"
    for name in table:
        assert name in names, f"{name} bound in Python but not declared in {header}"


def test_struct_sizes_match_the_reference():
    # renderer/gputypes.hpp:36, :72, :90, :115
    assert C.sizeof(abi.CameraPacked) == 416
    assert C.sizeof(abi.AtmospherePacked) == 128
    assert C.sizeof(abi.DirectionalLightPacked) == 176
    assert C.sizeof(abi.SpotLightPacked) == 192
    assert C.sizeof(abi.VertexPacked) == 48  # gputypes.hpp:126
    # std430 offsets used by the shaders (types/atmosphere.glinl)
    assert abi.AtmospherePacked.incidentDirectionSun.offset == 64
    assert abi.AtmospherePacked.sunIntensitySpectrum.offset == 112
    assert abi.AtmospherePacked.sunAngularRadius.offset == 124
    assert abi.CameraPacked.inverseProjection.offset == 64
    assert abi.CameraPacked.rotation.offset == 256
    assert abi.CameraPacked.position.offset == 400
    assert abi.SpotLightPacked.position.offset == 160
    assert abi.SpotLightPacked.falloffDistance.offset == 184


def test_abi_version():
    assert lib().szg_abi_version() == abi.SZG_ABI_VERSION


def test_rowtile_local_rows_partition_the_frame():
    f = lib().szg_rowtile_local_rows
    for height in (1, 7, 100, 1080, 2160, 4320):
        for block in (1, 5, 8, 16):
            for nranks in (1, 2, 3, 4, 8):
                rows = [f(height, block, r, nranks) for r in range(nranks)]
                assert sum(rows) == height, (height, block, nranks, rows)
                assert max(rows) - min(rows) <= block
    assert f(100, 0, 0, 2) == 0 and f(100, 8, 2, 2) == 0


def test_no_cpu_fallback_without_a_device():
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    handle = C.c_void_p()
    desc = abi.SkyviewDesc(512, 128, 2048, 1024, 0, 0)
    status = lib().szg_skyview_create(C.byref(handle), C.byref(desc), 0)
    assert status == -2 and not handle.value
    assert b"no CPU fallback" in lib().szg_last_error()
    ddesc = abi.DeferredDesc(64, 64, 4, 4, 0, 0)
    assert lib().szg_deferred_create(C.byref(handle), C.byref(ddesc), 0) == -2
    from syzygy_amd import pipelines

    assert pipelines.SkyViewComputePipeline.create() is None  # skyview.cpp:713-740: nullptr on failure


def test_null_arguments_are_rejected():
    assert lib().szg_skyview_create(None, None, 0) == -1
    assert lib().szg_deferred_create(None, None, 0) == -1
    assert lib().szg_compose_rowtiles(None, None, 0, 1, 1, None, 0, 0) == -1
    lib().szg_skyview_destroy(None)
    lib().szg_deferred_destroy(None)
    # every record entry point rejects a NULL pipeline before touching a device, and says why
    rect = abi.Rect(0, 0, 8, 8)
    assert lib().szg_deferred_record_gbuffer_raster(None, None, rect, None, None, 0, None, None, 0) == -1
    assert b"szg_deferred_record_gbuffer_raster" in lib().szg_last_error()
    assert lib().szg_deferred_record_shadow_raster(None, None, None, 0, None, 0, None, 0) == -1
    assert lib().szg_deferred_record_draw_commands_meshes(None, None, rect, None, None, 0, None, 0, None, 0, 0, None, None, 0) == -1
    assert lib().szg_deferred_record_lights(None, None, rect, None, None, 0, None, 0, None, 0, 0, None) == -1
    assert lib().szg_skyview_record_transmittance(None, None, 0, None) == -1
    assert lib().szg_record_oetf(None, None, 8, 8, abi.SZG_OETF_SRGB) == -1
    assert b"szg_record_oetf" in lib().szg_last_error()


def test_round2_entry_points_reject_bad_arguments_without_a_device():
    """The multi-GPU collectives, LUT reuse and row-slice entry points: NULL / inconsistent arguments are refused before
    anything touches a device (and before RCCL is loaded)."""
    L = lib()
    assert L.szg_rowtile_comm_unique_id(None) == -1
    handle = C.c_void_p()
    blob = C.create_string_buffer(abi.SZG_ROWTILE_COMM_ID_BYTES)
    assert L.szg_rowtile_comm_create(None, 0, 1, blob, 0) == -1
    assert L.szg_rowtile_comm_create(C.byref(handle), 2, 2, blob, 0) == -1 and not handle.value  # rank outside [0, nranks)
    assert L.szg_rowtile_comm_create(C.byref(handle), 0, 1, None, 0) == -1
    assert L.szg_rowtile_gather(None, None, None, 0, None, 0) == -1
    assert L.szg_rowtile_allgather(None, None, None, 0) == -1
    assert L.szg_rowtile_comm_rank(None) == -1 and L.szg_rowtile_comm_size(None) == -1
    L.szg_rowtile_comm_destroy(None)
    assert L.szg_skyview_set_lut_reuse(None, 1) == -1
    assert L.szg_skyview_invalidate_luts(None, abi.SZG_LUT_SKYVIEW) == -1
    b, e = C.c_uint32(), C.c_uint32()
    assert L.szg_skyview_lut_row_slice(None, 0, 1, C.byref(b), C.byref(e)) == -1
    assert L.szg_skyview_allgather_lut_rows(None, None, None) == -1


def test_communicator_needs_a_device():
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    handle = C.c_void_p()
    blob = C.create_string_buffer(abi.SZG_ROWTILE_COMM_ID_BYTES)
    # no GPU here: either RCCL cannot be loaded or there is no HIP device; never a communicator, never a CPU stand-in
    status = lib().szg_rowtile_comm_create(C.byref(handle), 0, 1, blob, 0)
    assert status == -2 and not handle.value


def test_host_raster_helpers_need_no_device():
    """szg_transform_matrix / szg_tick_mesh_instance / szg_calculate_shadow_bounds are CPU-only (include/szg/host.h)."""
    m = abi.Mat4()
    lib().szg_transform_matrix(abi.f3(1, 2, 3), abi.f3(0, 0, 0), abi.f3(2, 2, 2), C.byref(m))
    a = m.to_numpy()
    assert a[0, 0] == 2.0 and a[0, 3] == 1.0 and a[1, 3] == 2.0 and a[2, 3] == 3.0 and a[3, 3] == 1.0


def test_collectives_bind_the_rccl_that_is_already_in_the_process():
    """Inside a PyTorch process the C-ABI's collectives must use the RCCL torch.distributed uses (it is already mapped), never a
    second copy loaded by SONAME from /opt/rocm; SZG_RCCL_LIBRARY is the explicit override (szg_comm.cpp)."""
    import torch  # noqa: F401  (maps torch/lib/librccl.so)

    info = lib().szg_rowtile_comm_backend().decode()
    if "SZG_RCCL_LIBRARY" in os.environ:
        assert os.environ["SZG_RCCL_LIBRARY"] in info
    else:
        assert "already mapped in this process" in info and "librccl" in info and os.path.dirname(torch.__file__) in info, info
    assert "RCCL version code" in info
