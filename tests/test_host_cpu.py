"""CPU-only checks: host-side logic, the C-ABI library (load + exported symbols, host helpers),
and the world_size-2 sharding path over gloo.  No GPU compute."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, golden_rows
import dmpfold_oracle as O
from dmpfold2_amd import _lib, predict, shard, synth


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "dmpfold_hip.h")).read()
    declared = set(re.findall(r"\b(dmp_[a-z0-9_]+)\s*\(", header))
    declared -= {"dmp_ctx", "dmp_lane", "dmp_status", "dmp_pipeline"}
    assert len(declared) == 65                           # 49 of round 5 + the 16 dmp_pipeline_* entry points (round 6, ABI 5)
    lib = C.CDLL(_lib.LIB_PATH)
    missing = [name for name in sorted(declared) if not hasattr(lib, name)]
    assert not missing, missing
    # the ctypes binding covers the same set
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.load().dmp_abi_version() == _lib.ABI_VERSION == 5


def test_residue_encoding_all_bytes():
    """dmp_msa_encode (host helper of the C ABI) == the reference's translate + uint8 wrap
    (predict.py:124-128) for every byte value."""
    lib = _lib.load()
    text = np.arange(256, dtype=np.uint8)
    out = np.empty_like(text)
    assert lib.dmp_msa_encode(text.ctypes.data, text.size, out.ctypes.data) == 0
    assert np.array_equal(out, O._code_table()[text])
    assert out[ord("A")] == 0 and out[ord("V")] == 19 and out[ord("X")] == 20 and out[ord("-")] == 21
    assert out[ord("a")] == (ord("a") - 65) % 256


@pytest.mark.parametrize("name", ["pf10963_n0_m0", "alphabet_L16_N12_n0_m0", "synth_L24_N3050_n1_m0"])
def test_host_parse_and_encode_match_reference(name, tmp_path):
    g = load_golden(name)
    p = tmp_path / "a.aln"
    p.write_text(">header lines are skipped\n" + "\n".join(golden_rows(g)) + "  \n")
    alnmat = predict.encode_aln(predict.read_aln(str(p)))
    assert alnmat.dtype == np.uint8
    assert np.array_equal(alnmat, g["alnmat"])          # includes the 3000-row cap


def test_ragged_alignment_raises_value_error(tmp_path):
    p = tmp_path / "r.aln"
    p.write_text("ACDEFGHIKL\nACDEFGHIK\n")
    with pytest.raises(ValueError):
        predict.encode_aln(predict.read_aln(str(p)))


def test_template_parser_matches_oracle(tmp_path):
    g = load_golden("template_L96_N50_n1_m0")
    p = tmp_path / "t.pdb"
    with open(p, "w") as fh:
        fh.write("HEADER test\n")
        for i, (x, y, z) in enumerate(g["template_ca"]):
            fh.write("ATOM  %5d  N   ALA A%4d    %8.3f%8.3f%8.3f  1.00  0.00\n" % (2 * i, i + 1, x + 1, y, z))
            fh.write("ATOM  %5d  CA  ALA A%4d    %8.3f%8.3f%8.3f  1.00  0.00\n" % (2 * i + 1, i + 1, x, y, z))
    got = predict.read_template_ca(str(p))
    assert got.shape == (96, 3) and got.dtype == np.float32
    assert np.array_equal(got, O.read_template_ca(str(p)).numpy())


def test_pdb_text_matches_reference_cli_output():
    g = load_golden("pf10963_default_cli")
    text = predict.pdb_text(torch.from_numpy(g["coords"]), torch.from_numpy(g["confs"]), g["alnmat"])
    assert text == bytes(g["cli_stdout"]).decode()


def test_pdb_text_rejects_non_standard_first_row():
    coords, confs = torch.zeros(8, 5, 3), torch.zeros(8)
    alnmat = np.zeros((1, 8), dtype=np.uint8)
    alnmat[0, 3] = 20                                     # 'X' in the query: reference raises KeyError
    with pytest.raises(KeyError):
        predict.pdb_text(coords, confs, alnmat)


def test_no_cpu_path_and_no_download(tmp_path):
    with pytest.raises(RuntimeError):
        predict.aln_to_coords(os.path.join(ROOT, "tests", "golden", "PF10963.aln"), device="cpu")
    if not os.path.isfile(predict.default_weight_files()[0]):
        with pytest.raises(FileNotFoundError):
            predict.load_state_dict(None)


def test_context_rejects_bad_arguments_without_a_gpu():
    lib = _lib.load()
    ctx = C.c_void_p()
    assert lib.dmp_ctx_create(0, 4, 10, C.byref(ctx)) < 0          # max_L < 8
    assert b"max_L" in lib.dmp_last_error()
    assert lib.dmp_msa_encode(None, 0, None) < 0


def test_synthetic_generators_are_deterministic():
    a = synth.synth_msa(50, 20, seed=3)
    assert a == synth.synth_msa(50, 20, seed=3) and a != synth.synth_msa(50, 20, seed=4)
    assert len(a) == 20 and all(len(r) == 50 for r in a) and "-" not in a[0]
    sd = synth.synth_weights(0, coord_scale=5.0)
    assert len(sd) == 184 and sum(v.size for v in sd.values()) == 34956278
    assert np.array_equal(sd["embed.weight"], np.eye(22, dtype=np.float32))


def test_partition_targets_is_balanced_and_complete():
    rng = np.random.default_rng(0)
    Ls = rng.integers(100, 301, size=256)
    costs = [shard.estimate_cost(int(L), 2000) for L in Ls]
    for world in (1, 2, 4, 8):
        parts = shard.partition_targets(costs, world)
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(256))
        loads = [sum(costs[i] for i in p) for p in parts]
        assert max(loads) / (sum(loads) / world) < 1.02


_WORKER = r"""
import os, sys, time
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from dmpfold2_amd import shard
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
costs = [shard.estimate_cost(100 + 7 * i, 500) for i in range(23)]
mine = shard.partition_targets(costs, world)[rank]
dist.barrier()
t0 = time.perf_counter()
done = 0
for i in mine:                      # stand-in for the per-target GPU work: no collective in here
    done += 1
elapsed = time.perf_counter() - t0 + 0.01 * (rank + 1)
total, tmax = shard.job_summary(done, elapsed)
assert total == 23, total
assert abs(tmax - max(0.01 * (r + 1) for r in range(world))) < 0.05
gathered = [None] * world
dist.all_gather_object(gathered, mine)
assert sorted(i for p in gathered for i in p) == list(range(23))
# a rank that broke down (rank 1) and a rank with two failed targets (rank 0): EVERY rank's summary shows both
t4 = shard.job_summary(done if rank == 0 else 0, elapsed, failures=(2, 0) if rank == 0 else (0, 1))
assert t4[2] == 2 and t4[3] == 1 and t4[0] == len(gathered[0]), t4
dist.destroy_process_group()
print("rank", rank, "ok", len(mine))
"""


def test_sharded_job_over_gloo_world_size_2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


def test_batch_target_list_and_shards(tmp_path):
    """Batch front end, host logic: the targets file parser and that the per-rank shards of a job
    cover every target exactly once for 1..8 ranks."""
    from dmpfold2_amd import shard
    from dmpfold2_amd.batch import read_target_list
    lst = tmp_path / "t.txt"
    lst.write_text("a.aln\n\n# comment\nb.aln tpl.pdb  # trailing\n  c.aln\n")
    assert read_target_list(str(lst)) == [("a.aln", None), ("b.aln", "tpl.pdb"), ("c.aln", None)]
    costs = [shard.estimate_cost(L, N) for L, N in [(300, 2000), (82, 252), (500, 5000), (64, 3), (1000, 2000)] * 5]
    for world in range(1, 9):
        parts = shard.partition_targets(costs, world)
        assert sorted(i for p in parts for i in p) == list(range(len(costs)))


def test_read_a3m_matches_readme_recipe(tmp_path):
    """egrep -v "^>" x.a3m | sed 's/[a-z]//g' (reference README.md:30-33)."""
    from dmpfold2_amd.predict import read_a3m, read_aln
    a3m = tmp_path / "x.a3m"
    a3m.write_text(">query\nACDEFGHIKL\n>hit1 desc\nACd-EFGHikIKL\n>hit2\n-CDEFGHIK-\n")
    aln = tmp_path / "x.aln"
    aln.write_text("ACDEFGHIKL\nAC-EFGHIKL\n-CDEFGHIK-\n")
    assert read_a3m(str(a3m)) == read_aln(str(aln))


def test_no_hazardous_packed_f32_instruction_in_any_kernel():
    """tools/isa_lint.py: no kernel of the library contains v_pk_{mul,add,fma}_f32 with op_sel[1] = 1, the form
    that returns 0 in lanes 48..63 beside f16 MFMA waves on MI355X.  The rule itself is
    checked on the instructions measured by tools/pk_hazard.hip."""
    import importlib.util
    import shutil
    spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "tools", "isa_lint.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    assert lint.hazardous("\tv_pk_mul_f32 v[0:1], v[4:5], v[0:1] op_sel:[0,1] op_sel_hi:[1,0]")
    assert lint.hazardous("\tv_pk_add_f32 v[0:1], s[2:3], v[0:1] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]")
    assert lint.hazardous("\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0] op_sel_hi:[1,0,1]")
    assert not lint.hazardous("\tv_pk_mul_f32 v[0:1], v[4:5], v[2:3] op_sel:[1,0] op_sel_hi:[0,1]")
    assert not lint.hazardous("\tv_pk_mul_f32 v[4:5], v[10:11], v[2:3] op_sel_hi:[1,0]")
    assert not lint.hazardous("\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,0,1] op_sel_hi:[1,1,0]")
    assert not lint.hazardous("\tv_pk_mov_b32 v[2:3], v[6:7], v[6:7] op_sel:[1,0]")
    # second rule (round 5): a VALU write of an operand of an INLINE-ASSEMBLY MFMA fewer than two wait states ahead of it
    # (the compiler does not see the MFMA as a reader: found on the GPU in the first build of vgru_f32.hip)
    asm = lambda body: "kern:\n" + body                  # noqa: E731
    mfma = "\t;;#ASMSTART\n\tv_mfma_f32_16x16x4_f32 v[98:101], a16, v62, v[98:101]\n\t;;#ASMEND\n"
    assert lint.scan_asm_mfma_hazards(asm("\tv_mov_b64_e32 v[98:99], s[28:29]\n" + mfma))            # the measured failure
    assert lint.scan_asm_mfma_hazards(asm("\tv_mov_b64_e32 v[98:99], s[28:29]\n\tv_mov_b32_e32 v7, v8\n" + mfma))   # one slot
    assert lint.scan_asm_mfma_hazards(asm("\tv_mov_b32_e32 v62, v8\n" + mfma))                          # srcB
    assert not lint.scan_asm_mfma_hazards(asm("\tv_mov_b64_e32 v[98:99], s[28:29]\n\ts_nop 1\n" + mfma))   # two wait states
    assert not lint.scan_asm_mfma_hazards(asm("\tv_mov_b64_e32 v[98:99], s[28:29]\n\t;;#ASMSTART\n\ts_nop 1\n"
                                              "\tv_mfma_f32_16x16x4_f32 v[98:101], a16, v62, v[98:101]\n\t;;#ASMEND\n"))
    assert not lint.scan_asm_mfma_hazards(asm("\tv_mov_b64_e32 v[90:91], s[28:29]\n" + mfma))           # another register
    assert not lint.scan_asm_mfma_hazards(asm("\tv_mov_b64_e32 v[98:99], s[28:29]\n"                    # a builtin MFMA: the
                                              "\tv_mfma_f32_16x16x4_f32 v[98:101], a16, v62, v[98:101]\n"))  # compiler's own business
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    assert lint.main() == 0


# ---------------------------------------------------------------------------- round 2
def test_unknown_residue_letter_raises_index_error(tmp_path):
    """A character outside the alignment alphabet maps to a code above 21; the reference fails in its
    22-row embedding (network.py:223) with IndexError.  Raised on the host before anything runs; rows
    beyond the 3000-row cap are never looked at, as in the reference."""
    p = tmp_path / "bad.aln"
    p.write_text("ACDEFGHIKL\nACDEFgHIKL\n")
    with pytest.raises(IndexError):
        predict.encode_aln(predict.read_aln(str(p)))
    rows = ["ACDEFGHIKL"] * 3000 + ["ACDEF1HIKL"]
    assert predict.encode_aln(rows).shape == (3000, 10)
    assert predict.encode_aln(["ACDEFGHIKL", "BJOUXZ-.AC"]).max() == 21


def test_fault_bits_map_to_the_reference_exceptions():
    predict.raise_for_faults(0)
    with pytest.raises(IndexError):
        predict.raise_for_faults(predict.FAULT_BAD_CODE | predict.FAULT_F16_RANGE)
    with pytest.raises(predict.DeviceFault) as ei:
        predict.raise_for_faults(predict.FAULT_F16_RANGE | predict.FAULT_SEQ_HANDOFF)
    assert ei.value.bits == 3 and isinstance(ei.value, RuntimeError)
    header = open(os.path.join(ROOT, "include", "dmpfold_hip.h")).read()
    for name, bit in (("SEQ_HANDOFF", 1), ("F16_RANGE", 2), ("REFINE_HANDOFF", 4), ("BAD_CODE", 8)):
        assert re.search(r"#define DMP_FAULT_%s %d\b" % (name, bit), header)
        assert getattr(predict, "FAULT_" + name) == bit


def test_two_part_default_weights_are_merged(tmp_path, monkeypatch, synth_sd):
    """predict.py:81-96: the default model is two pickled dicts merged with dict.update; a single
    file through weights_file= gives the same state_dict."""
    keys = list(synth_sd)
    sd = {k: torch.from_numpy(np.array(v)) for k, v in synth_sd.items()}
    d = tmp_path / "trained_model"
    d.mkdir()
    parts = [str(d / f"FINAL_fullmap_e2e_model_part{i}.pt") for i in (1, 2)]
    torch.save({k: sd[k] for k in keys[:90]}, parts[0])
    torch.save({k: sd[k] for k in keys[90:]}, parts[1])
    single = str(tmp_path / "one.pt")
    torch.save(sd, single)
    monkeypatch.setattr(predict, "default_weight_files", lambda: parts)
    merged = predict.load_state_dict(None)
    one = predict.load_state_dict(single)
    assert list(merged) == keys and list(one) == keys
    assert all(torch.equal(merged[k], one[k]) for k in keys)


def test_weight_files_are_loaded_as_data_only(tmp_path):
    """A -w file is a tensor dict; anything that needs unpickling code is refused."""
    import pickle

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    p = tmp_path / "evil.pt"
    torch.save({"embed.weight": Evil()}, str(p))
    with pytest.raises(pickle.UnpicklingError):
        predict.load_state_dict(str(p))


def test_bench_launches_its_own_ranks_over_gloo():
    """`python bench.py --gpus 2` outside a launcher starts two ranks itself (the launch, barrier and
    max-over-ranks code of bench.py with a stand-in workload, gloo backend)."""
    env = dict(os.environ, DMP_BENCH_STUB="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    import json
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["steps"] == 2
    # slowest rank sleeps 0.02 * 2 * steps: the reported time is the maximum over the ranks
    assert out["ms_per_step"] >= 39.0
    # under a launcher (RANK set) the same command line must not launch again
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "1"], 29555)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "1"]


def test_north_star_golden_input_is_reproducible():
    """The L=300, N=2000 golden stores only a checksum of its alignment: the generator must give the
    same bytes here and on the GPU box."""
    import hashlib
    g = load_golden("synth_L300_N2000_n1_m0")
    alnmat = predict.encode_aln(synth.synth_msa(300, 2000, int(g["msa_seed"])))
    assert hashlib.sha256(alnmat.tobytes()).hexdigest() == bytes(g["alnmat_sha256"]).decode()
    assert g["coords"].shape == (300, 5, 3) and g["ca_pass"].shape == (2, 300, 3)
    assert float(g["oracle_vs_ref_ca_rmsd"]) <= 1e-3 and float(g["oracle_vs_ref_conf"]) < 1e-4
    g10 = load_golden("synth_L300_N2000_n10_m0")
    assert bytes(g10["alnmat_sha256"]) == bytes(g["alnmat_sha256"]) and g10["ca_pass"].shape == (11, 300, 3)
    assert float(g10["oracle_vs_ref_ca_rmsd"]) <= 1e-3 and float(g10["oracle_vs_ref_conf"]) < 1e-4
    assert np.abs(g10["ca_pass"][:2] - g["ca_pass"]).max() == 0.0      # the same first two passes


def test_batch_reports_bad_targets_without_losing_the_others(tmp_path):
    """Host half of the per-target failure path: an alignment with an unknown letter is reported as
    that target's failure (BatchFailures), before any GPU is needed."""
    from dmpfold2_amd.batch import BatchFailures, run_batch
    bad = tmp_path / "bad.aln"
    bad.write_text("ACDEFGHIKL\nACDEFgHIKL\n")
    with pytest.raises(BatchFailures) as ei:
        run_batch([(str(bad), None)], str(tmp_path / "out"), 0, 0, state_dict={})
    assert len(ei.value.failed) == 1 and isinstance(ei.value.failed[0][1], IndexError)
    assert ei.value.outputs == []


def test_batch_inputs_and_output_variants(tmp_path):
    """Batch front end, host side: directories expand to their alignments in sorted order; the `ca` variant is
    the CLI's text reduced to its CA records; `npz` round-trips the arrays."""
    from dmpfold2_amd.batch import ca_only_text, expand_inputs, write_result
    d = tmp_path / "msas"
    d.mkdir()
    for name in ("b.aln", "a.a3m", "notes.txt", "c.aln"):
        (d / name).write_text("ACDEFGHIKL\n")
    lone = tmp_path / "x.aln"
    lone.write_text("ACDEFGHIKL\n")
    got = expand_inputs([str(d), str(lone)])
    assert [os.path.basename(a) for a, _ in got] == ["a.a3m", "b.aln", "c.aln", "x.aln"] and all(t is None for _, t in got)
    # the README workflow leaves x.a3m beside its x.aln: only the .aln is a target (both would be written to x.pdb)
    (d / "a.aln").write_text("ACDEFGHIKL\n")
    assert [os.path.basename(a) for a, _ in expand_inputs([str(d)])] == ["a.aln", "b.aln", "c.aln"]
    from dmpfold2_amd.batch import check_output_stems
    check_output_stems(expand_inputs([str(d), str(lone)]))
    with pytest.raises(ValueError, match="both be written"):
        check_output_stems([(str(d / "b.aln"), None), (str(tmp_path / "other" / "b.a3m"), None)])
    g = load_golden("pf10963_default_cli")
    coords, confs = torch.from_numpy(g["coords"]), torch.from_numpy(g["confs"])
    full = bytes(g["cli_stdout"]).decode()
    ca = ca_only_text(coords, confs, g["alnmat"])
    lines = ca.split("\n")
    atoms = [ln for ln in lines if ln.startswith("ATOM")]
    assert len(atoms) == 82 and all(ln[12:16] == " CA " for ln in atoms)
    assert lines[0] == full.split("\n")[0] and lines[-2] == "END"
    ref_ca = [ln for ln in full.split("\n") if ln.startswith("ATOM") and ln[12:16] == " CA "]
    assert [ln[11:] for ln in atoms] == [ln[11:] for ln in ref_ca]            # same records, renumbered
    assert [int(ln[6:11]) for ln in atoms] == list(range(1, 83))
    out = tmp_path / "out"
    out.mkdir()
    assert open(write_result(str(out), "t/pf.aln", coords, confs, g["alnmat"], "pdb")).read() == full
    z = np.load(write_result(str(out), "t/pf.aln", coords, confs, g["alnmat"], "npz"))
    assert np.array_equal(z["coords"], g["coords"]) and np.array_equal(z["confs"], g["confs"])
    assert np.array_equal(z["alnmat"], g["alnmat"])
    from dmpfold2_amd.batch import run_batch
    with pytest.raises(ValueError):
        run_batch([], str(out), fmt="cif")


def test_empty_alignment_raises_index_error(tmp_path):
    """predict.py:124: `len(aln[0])` on an alignment without sequence lines (empty file, headers only)."""
    for text in ("", ">only a header\n"):
        p = tmp_path / "e.aln"
        p.write_text(text)
        with pytest.raises(IndexError):
            predict.encode_aln(predict.read_aln(str(p)))
        with pytest.raises(IndexError):
            O.encode_aln(O.read_aln(str(p)))


# ---------------------------------------------------------------------------- round 3
class _FakePipeline:
    """Stands in for the GPU scheduler in the host-logic tests of run_batch (its streaming interface: submit /
    step / poll / peek / result / collect): the 'prediction' of a target is a function of its alignment only, so
    results can be compared across shardings.  A target completes two scheduling rounds after its submission."""
    made = 0

    def __init__(self, device, max_L, max_N, state_dict, streams=2, stagger=False, precision=None):
        _FakePipeline.made += 1
        self.jobs, self.max_L, self.max_N = [], max_L, max_N
        self.age, self.ready, self.late, self.max_backlog = {}, {}, {}, 0

    def submit(self, d_msa, iterations, minsteps, template_ca=None):
        if d_msa.shape[1] > self.max_L or d_msa.shape[0] > self.max_N:
            raise RuntimeError("alignment exceeds the pipeline capacity")
        self.jobs.append(d_msa)
        t = len(self.jobs) - 1
        self.age[t] = 0
        self.max_backlog = max(self.max_backlog, len(self.age))
        return t

    def _value(self, t):
        m = self.jobs[t]
        L = m.shape[1]
        base = m.float().mean() + torch.arange(L * 15, dtype=torch.float32).reshape(L, 5, 3) * 0.01
        return base, torch.full((L,), float(m.shape[0]) / 1000.0)

    def step(self):
        for t in list(self.age):
            self.age[t] += 1
            if self.age[t] >= 2:
                del self.age[t]
                self.late[t] = 3                  # issued to the end: completes on the "GPU" three polls later
        return True

    def backlog(self):
        return 0

    def busy(self):
        return bool(self.age)

    def poll(self):
        for t in list(self.late):
            self.late[t] -= 1
            if self.late[t] <= 0:
                del self.late[t]
                self.ready[t] = self._value(t)
        done = [t for t in self.ready if t not in getattr(self, "_polled", set())]
        self._polled = getattr(self, "_polled", set()) | set(done)
        return done

    def peek(self, t):
        return self.ready[t]

    def result(self, t):
        return self.ready.pop(t)

    def collect(self, tickets):
        return {t: self.ready.pop(t) if t in self.ready else self._value(t) for t in tickets}

    def close(self):
        pass


def _write_synth_targets(folder, count, seed=0):
    rng = np.random.default_rng(seed)
    paths = []
    for i in range(count):
        L, N = int(rng.integers(20, 61)), int(rng.integers(2, 40))
        p = folder / f"t{i:03d}.aln"
        synth.write_aln(str(p), synth.synth_msa(L, N, seed=1000 + i))
        paths.append((str(p), None))
    return paths


def test_run_batch_256_targets_world_8_each_exactly_once(tmp_path, monkeypatch):
    """BASELINE configs[3] at its own count on the host side: 256 targets, 8 ranks (run one after the other
    here).  Every target is written exactly once, by the rank that owns it; a rank reads and encodes ONLY its
    own targets (the partition comes from a header scan); results do not depend on the sharding."""
    from dmpfold2_amd import batch
    monkeypatch.setattr(batch, "Pipeline", _FakePipeline)
    d = tmp_path / "msas"
    d.mkdir()
    targets = _write_synth_targets(d, 256)
    reads = []
    real_read = batch.read_aln
    monkeypatch.setattr(batch, "read_aln", lambda p: (reads.append(p), real_read(p))[1])
    outs, owners = {}, {}
    for rank in range(8):
        before = len(reads)
        n, _, written = batch.run_batch(targets, str(tmp_path / f"out8"), 1, 0, state_dict={}, device="cpu",
                                        rank=rank, world=8)
        mine = batch.plan_shard(targets, 1, rank, 8)
        assert n == len(mine) == len(written) == len(reads) - before      # parsed its own shard, nothing else
        assert sorted(reads[before:]) == sorted(targets[i][0] for i in mine)
        for path in written:
            assert path not in outs
            outs[path] = open(path).read()
            owners[path] = rank
    assert len(outs) == 256 and len(set(reads)) == 256 and len(reads) == 256
    loads = np.bincount(list(owners.values()), minlength=8)
    assert loads.min() >= 16                                              # no rank starves
    # one rank alone writes the same bytes
    n, _, written = batch.run_batch(targets, str(tmp_path / "out1"), 1, 0, state_dict={}, device="cpu")
    assert n == 256
    for path in written:
        assert open(path).read() == outs[os.path.join(str(tmp_path / "out8"), os.path.basename(path))]


def test_scan_target_estimates_without_parsing(tmp_path):
    from dmpfold2_amd.batch import scan_target
    p = tmp_path / "x.aln"
    synth.write_aln(str(p), synth.synth_msa(57, 123, 4))
    assert scan_target(str(p)) == (57, 123)
    a3m = tmp_path / "y.a3m"
    a3m.write_text(">q\nACDEFGHIKL\n>h1\nACdeDEFGHIKL\n>h2\nACDEFGHIKL\n")
    assert scan_target(str(a3m)) == (10, 3)
    assert scan_target(str(tmp_path / "missing.aln")) == (0, 0)
    # the count must not depend on a final newline or on blanks behind the first row (a size / line-length
    # estimate came out short - 49 and 42 for 50 rows - and the engines were built too small for that target)
    rows = synth.synth_msa(31, 50, 9)
    q = tmp_path / "no_newline.aln"
    q.write_text("\n".join(rows))
    assert scan_target(str(q)) == (31, 50)
    r = tmp_path / "blanks.aln"
    r.write_text(rows[0] + "      \n" + "\n".join(rows[1:]) + "\n")
    assert scan_target(str(r)) == (31, 50)
    h = tmp_path / "headers.aln"
    h.write_text(">first\n" + "\n>x\n".join(rows) + "\n")
    assert scan_target(str(h)) == (31, 50)


def test_batch_engines_are_sized_for_a_file_without_final_newline(tmp_path, monkeypatch):
    """run_batch sizes the engines from the scans; the deepest target of a rank, written without
    a final newline, must still fit (it was reported as failed: 'alignment exceeds the pipeline capacity')."""
    from dmpfold2_amd import batch
    made = {}

    class Sized(_FakePipeline):                          # _FakePipeline.submit refuses what exceeds (max_L, max_N)
        def __init__(self, device, max_L, max_N, sd, streams=4, precision=None):
            made["cap"] = (max_L, max_N)
            super().__init__(device, max_L, max_N, sd, streams=streams)

    monkeypatch.setattr(batch, "Pipeline", Sized)
    rows = synth.synth_msa(24, 50, 3)
    deep = tmp_path / "deep.aln"
    deep.write_text("\n".join(rows))                    # no final newline
    small = tmp_path / "small.aln"
    synth.write_aln(str(small), synth.synth_msa(20, 7, 4))
    n, _, written = batch.run_batch([(str(small), None), (str(deep), None)], str(tmp_path / "out"), 0, 0,
                                    state_dict={}, device="cpu")
    assert n == 2 and len(written) == 2 and made["cap"] == (24, 50)


def test_batch_unreadable_file_fails_that_target_only(tmp_path, monkeypatch):
    """a missing / undecodable alignment is that target's failure, not the shard's."""
    from dmpfold2_amd import batch
    monkeypatch.setattr(batch, "Pipeline", _FakePipeline)
    good = tmp_path / "good.aln"
    synth.write_aln(str(good), synth.synth_msa(24, 5, 1))
    binary = tmp_path / "binary.aln"
    binary.write_bytes(b"ACDE\xff\xfeFGH\nACDEFGHIK\n")
    with pytest.raises(batch.BatchFailures) as ei:
        batch.run_batch([(str(tmp_path / "nope.aln"), None), (str(good), None), (str(binary), None)],
                        str(tmp_path / "out"), 0, 0, state_dict={}, device="cpu")
    kinds = {os.path.basename(a): type(e) for a, e in ei.value.failed}
    assert kinds["nope.aln"] is FileNotFoundError and issubclass(kinds["binary.aln"], (UnicodeDecodeError, ValueError, IndexError))
    assert [os.path.basename(p) for p in ei.value.outputs] == ["good.pdb"]


def test_bench_stub_world_8_over_gloo():
    """The driver's 8-rank command line (torch.distributed.run, one process per GPU) through bench.py's launch,
    barrier and max-over-ranks code with the stand-in workload: ONE JSON line, n_gpus = ranks = 8."""
    import json
    env = dict(os.environ, DMP_BENCH_STUB="1", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["ranks"] == 8
    assert out["ms_per_step"] >= 0.02 * 8 * 1e3 * 0.95           # the slowest rank (rank 7) sets the time


def test_length_limit_is_reported_before_anything_runs(tmp_path):
    """The reference has no length cap; this build's is DMP_MAX_L = 2048 (include/dmpfold_hip.h): the drop-in raises
    RuntimeError naming the length before touching the device, and dmp_ctx_create refuses the size."""
    import ctypes as C
    from dmpfold2_amd import _lib, predict
    assert predict.MAX_L == 2048
    with pytest.raises(RuntimeError, match="2049 columns"):
        predict.get_engine("cuda:0", 2049, 4, state_dict={})
    lib = _lib.load()
    ctx = C.c_void_p()
    assert lib.dmp_ctx_create(0, 2049, 4, C.byref(ctx)) != 0
    assert b"2048" in lib.dmp_last_error()


_QUEUE_WORKER = r"""
import os, sys, time
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r}); sys.path.insert(0, os.path.join({root!r}, "oracle"))
import torch, torch.distributed as dist
from dmpfold2_amd import batch
import test_host_cpu as T
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
from dmpfold2_amd import shard
store = shard.job_store(rank, world)            # public constructor; the same store carries the process group
dist.init_process_group("gloo", store=store, rank=rank, world_size=world)
batch.Pipeline = T._FakePipeline
if rank == 1:                                   # a slow rank: the others take what it does not get to
    real = T._FakePipeline.step
    T._FakePipeline.step = lambda self: (time.sleep(0.01), real(self))[1]
targets = [l.split()[0] for l in open({listfile!r})]
targets = [(t, None) for t in targets]
n, secs, outs = batch.run_batch(targets, {out!r}, 1, 0, state_dict={{}}, device="cpu", rank=rank, world=world, store=store)
gathered = [None] * world
dist.all_gather_object(gathered, sorted(os.path.basename(o) for o in outs))
names = sorted(x for g in gathered for x in g)
assert names == sorted(os.path.splitext(os.path.basename(t))[0] + ".pdb" for t, _ in targets), (len(names), len(targets))
assert all(len(g) > 0 for g in gathered)
if rank == 0:
    print("taken per rank:", [len(g) for g in gathered])
# ADVICE r03: a SECOND job over the same store starts its own counter (it used to find the first one's exhausted and
# return (0, 0.0, []) without an error)
dist.barrier()
n2, _, outs2 = batch.run_batch(targets, {out!r} + "_again", 1, 0, state_dict={{}}, device="cpu", rank=rank, world=world, store=store)
again = [None] * world
dist.all_gather_object(again, len(outs2))
assert sum(again) == len(targets), again
dist.destroy_process_group()
print("rank", rank, "ok", n)
"""


def test_shared_work_queue_over_gloo_world_size_2(tmp_path):
    """Several ranks, one queue: every rank takes the next most expensive target from an atomic counter in the job's
    key-value store whenever it has room (no collective, nothing on the data path).  Two gloo ranks, one of them slow:
    every target is predicted exactly once and the fast rank takes more of them."""
    d = tmp_path / "msas"
    d.mkdir()
    targets = _write_synth_targets(d, 40)
    (tmp_path / "list.txt").write_text("".join(a + "\n" for a, _ in targets))
    script = tmp_path / "qworker.py"
    script.write_text(_QUEUE_WORKER.format(root=ROOT, tests=os.path.join(ROOT, "tests"),
                                           listfile=str(tmp_path / "list.txt"), out=str(tmp_path / "out")))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29531", str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2
    taken = eval(r.stdout.split("taken per rank:")[1].splitlines()[0])
    assert sum(taken) == 40 and taken[0] > taken[1]
    assert len(os.listdir(tmp_path / "out")) == 40


class _FakeStore:
    """torch.distributed.Store.add semantics for the host-logic tests"""

    def __init__(self):
        self.kv = {}

    def add(self, key, n):
        self.kv[key] = self.kv.get(key, 0) + n
        return self.kv[key]


def test_shared_queue_exhausted_at_first_take_is_not_an_error():
    """8 ranks and 5 targets - or a rank that starts after the others took everything - find the
    counter past the end at their FIRST take.  That rank has nothing to do; it must not raise (it would skip the job's
    summary reduction and hang the other ranks)."""
    from dmpfold2_amd import batch
    store = _FakeStore()
    queues = [batch._SharedQueue(list(range(5)), store, "job") for _ in range(8)]
    batch._SharedQueue._calls.pop("job", None)         # the eight objects stand for eight processes' first queue
    for q in queues:
        q._key = "dmpfold_batch_next/job/0"
    got = [q.take() for q in queues]
    assert sorted(x for x in got if x is not None) == [0, 1, 2, 3, 4] and got[5:] == [None, None, None]
    assert all(q.take() is None for q in queues)       # and stays exhausted, quietly


def test_run_batch_rank_without_targets_returns_empty(tmp_path):
    """the same through run_batch: the late rank returns (0, 0.0, [])-like counts instead of raising"""
    from dmpfold2_amd import batch
    d = tmp_path / "msas"
    d.mkdir()
    targets = _write_synth_targets(d, 3)
    store = _FakeStore()
    real = batch.Pipeline
    batch.Pipeline = _FakePipeline
    try:
        n0, _, outs0 = batch.run_batch(targets, str(tmp_path / "o"), 1, 0, state_dict={}, device="cpu", rank=0, world=2,
                                       store=store)
        batch._SharedQueue._calls.clear()              # rank 1 is another process: its first queue over this list
        n1, _, outs1 = batch.run_batch(targets, str(tmp_path / "o"), 1, 0, state_dict={}, device="cpu", rank=1, world=2,
                                       store=store)
    finally:
        batch.Pipeline = real
    assert n0 == 3 and len(outs0) == 3 and n1 == 0 and outs1 == []


def test_scan_target_counts_in_chunks(tmp_path):
    """scan_target reads fixed-size chunks; a header line that starts exactly at a chunk boundary, a missing
    final newline and CRLF blanks are counted as the whole-file scan counted them."""
    from dmpfold2_amd import batch
    row = b"ACDEFGHIKLMNPQRSTVWY" * 5                   # 100 columns
    for tail in (b"\n", b""):
        for pad in (0, 1, 2):
            # headers so long that a '>' lands on byte 1 << 20 exactly for one of the pads
            body = b""
            n_rows = 0
            while len(body) < (1 << 20) - 200:
                body += b">s%d\n" % n_rows + row + b"\n"
                n_rows += 1
            fill = (1 << 20) - len(body) - 1 + pad
            body += b">" + b"x" * (fill - 2) + b"\n"   # a header that ends right at / around the boundary
            body += b">next\n" + row + b"\n>last\n" + row + tail
            n_rows += 3
            p = tmp_path / f"t{pad}{len(tail)}.aln"
            p.write_bytes(body)
            data = p.read_bytes()
            want = (data.count(b"\n") + (0 if data.endswith(b"\n") else 1)) - (data.count(b"\n>") + 1)
            assert batch.scan_target(str(p)) == (100, want), (pad, tail)
    a3m = tmp_path / "x.a3m"
    a3m.write_bytes(b">q\nACDaaEF-G\n>h\nAC-EFGG\n")
    assert batch.scan_target(str(a3m)) == (7, 2)
    assert batch.scan_target(str(tmp_path / "missing.aln")) == (0, 0)


def test_core_slices_follow_the_numa_node_of_each_gpu(tmp_path):
    """shard.plan_core_slices / gpu_local_cores on a made-up sysfs tree: 8 GPUs, four per NUMA node, 2 x 16 cores -
    every rank gets 4 cores of ITS GPU's node, disjoint; a GPU whose node is not stated (-1) is left alone."""
    from dmpfold2_amd import shard
    assert shard.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    root = tmp_path / "sys"
    bdfs = ["0000:%02x:00.0" % b for b in (0x05, 0x15, 0x65, 0x75, 0x85, 0x95, 0xe5, 0xf5)]
    for i, b in enumerate(bdfs):
        d = root / "bus/pci/devices" / b
        d.mkdir(parents=True)
        (d / "numa_node").write_text("%d\n" % (i // 4))
    for node, text in ((0, "0-15\n"), (1, "16-31\n")):
        d = root / "devices/system/node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(text)
    local = [shard.gpu_local_cores(b, str(root)) for b in bdfs]
    assert local[0] == list(range(16)) and local[7] == list(range(16, 32))
    plan = shard.plan_core_slices(local, range(32))
    assert plan[0] == [0, 1, 2, 3] and plan[3] == [12, 13, 14, 15] and plan[4] == [16, 17, 18, 19] and plan[7] == [28, 29, 30, 31]
    assert sorted(c for p in plan for c in p) == list(range(32))
    (root / "bus/pci/devices" / bdfs[2] / "numa_node").write_text("-1\n")
    local = [shard.gpu_local_cores(b, str(root)) for b in bdfs]
    plan = shard.plan_core_slices(local, range(32))
    assert plan[2] is None and plan[0] == [0, 1, 2, 3, 4] and plan[1] == [5, 6, 7, 8, 9]
    assert shard.gpu_local_cores("0000:aa:00.0", str(root)) == []          # no such device: unknown
    # restricted affinity: only allowed cores are dealt out; too few of them -> leave the ranks alone
    assert shard.plan_core_slices([list(range(16))] * 4, range(4)) == [None] * 4


def test_rank_core_slices_are_disjoint_and_cover_the_machine():
    """shard.pin_rank_to_cores: run in child processes (the affinity of the test process stays as it is)."""
    code = ("import os, sys; sys.path.insert(0, %r); from dmpfold2_amd import shard; "
            "print(shard.pin_rank_to_cores(int(sys.argv[1]), int(sys.argv[2])))" % ROOT)
    ncpu = os.cpu_count() or 1
    if len(os.sched_getaffinity(0)) != ncpu or ncpu < 4:
        pytest.skip("affinity already restricted or too few cores")
    world = 2
    env = dict(os.environ, DMP_PIN_CORES="1")
    got = [eval(subprocess.run([sys.executable, "-c", code, str(r), str(world)], capture_output=True, text=True,
                               check=True, env=env).stdout) for r in range(world)]
    off = eval(subprocess.run([sys.executable, "-c", code, "0", str(world)], capture_output=True, text=True,
                              check=True, env=dict(os.environ, DMP_PIN_CORES="0")).stdout)
    assert len(off) == ncpu                           # not opted in: placement is left to the operating system
    assert not set(got[0]) & set(got[1]) and len(got[0]) == len(got[1]) == ncpu // world
    one = eval(subprocess.run([sys.executable, "-c", code, "0", "1"], capture_output=True, text=True, check=True).stdout)
    assert len(one) == ncpu                           # a single rank is left alone
