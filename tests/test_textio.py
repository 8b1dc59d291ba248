"""The reference's text formats: loader semantics of project.cu:103-161, the writers, and the parser
contract of plot_quadtree.py:11-45 (checked on the REFERENCE's own output file, a golden fixture)."""
import os

import numpy as np
import pytest

from gpu_nbody_simulation_amd import textio

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
D = os.path.join(GOLD, "init1024")
FILES = [os.path.join(D, f"{n}_init.txt") for n in ("masses", "positions", "velocities")]


def test_loader_reads_first_n_lines(capsys):
    m, p, v = textio.loadSimulationDataFromText(*FILES, 100)
    assert "Loaded 100 bodies from text files." in capsys.readouterr().out      # project.cu:160
    assert m.shape == (100,) and p.shape == (100, 2) and v.shape == (100, 2)
    assert m[0] == 0.514535 and p[0].tolist() == [0.0790511, 0.0142126]
    assert v[2].tolist() == [8.96757e-05, 7.9058e-05]
    m2, p2, v2 = textio.loadSimulationDataFromText(*FILES, 1024, verbose=False)
    assert np.array_equal(m2[:100], m) and np.array_equal(p2, np.loadtxt(FILES[1]))


def test_loader_errors_match_the_reference(tmp_path):
    with pytest.raises(IndexError, match="Requested number of bodies exceeds N_BODIES."):
        textio.loadSimulationDataFromText(*FILES, 11, N_BODIES=10)
    with pytest.raises(RuntimeError, match="Failed to open file: nope.txt"):
        textio.loadSimulationDataFromText("nope.txt", FILES[1], FILES[2], 1)
    with pytest.raises(RuntimeError, match="Not enough mass entries in file"):
        textio.loadSimulationDataFromText(*FILES, 1025, verbose=False)
    short = tmp_path / "p.txt"
    short.write_text("0.1 0.2\n")
    with pytest.raises(RuntimeError, match="Not enough vector entries in file"):
        textio.loadSimulationDataFromText(FILES[0], str(short), FILES[2], 2, verbose=False)
    bad = tmp_path / "b.txt"
    bad.write_text("0.1\n0.3 0.4\n")
    with pytest.raises(RuntimeError, match="Failed to parse vector component in file"):
        textio.loadSimulationDataFromText(FILES[0], str(bad), FILES[2], 2, verbose=False)


def test_stod_prefix_semantics(tmp_path):
    f = tmp_path / "m.txt"
    f.write_text("  1.5e-3kg\n-2.\n.5 7\n")
    p = tmp_path / "p.txt"
    p.write_text("1 2\n3 4\n5 6\n")
    m, _, _ = textio.loadSimulationDataFromText(str(f), str(p), str(p), 3, verbose=False)
    assert m.tolist() == [1.5e-3, -2.0, 0.5]
    f.write_text("abc\n")
    with pytest.raises(ValueError):
        textio.loadSimulationDataFromText(str(f), str(p), str(p), 1, verbose=False)


def test_save_init_files_formats(tmp_path):
    m = np.array([0.514535, 12345.678, 1e-7])
    p = np.array([[0.0790511, 0.0142126], [1.0, -2.5e-5], [3.0, 4.0]])
    names = [str(tmp_path / n) for n in ("m.txt", "p.txt", "v.txt")]
    textio.save_init_files(m, p, p, *names)
    assert open(names[0]).read().split() == ["0.514535", "12345.7", "1e-07"]   # ostream << double
    assert open(names[1]).readline() == "0.0790511 0.0142126\n"
    textio.save_init_files(m, p, p, *names, exact=True)
    m2, p2, _ = textio.loadSimulationDataFromText(*names, 3, verbose=False)
    assert np.array_equal(m2, m) and np.array_equal(p2, p)


def test_shipped_files_round_trip_through_our_writer(tmp_path):
    """Writing what we loaded reproduces the reference's files byte for byte (6 significant digits)."""
    m, p, v = textio.loadSimulationDataFromText(*FILES, 1024, verbose=False)
    names = [str(tmp_path / n) for n in ("m.txt", "p.txt", "v.txt")]
    textio.save_init_files(m, p, v, *names)
    for ours, ref in zip(names, FILES):
        assert open(ours).read() == open(ref).read()


def test_save_positions_format(tmp_path):
    out = tmp_path / "positions.txt"
    textio.save_positions(str(out), [(0.0, np.array([[0.5, -0.25]])), (1.0, np.array([[1e-7, 2.0]]))])
    assert out.read_text() == "0.000000 0 0.500000 -0.250000 \n1.000000 0 0.000000 2.000000 \n"


def test_parser_on_the_reference_output(gold, tmp_path):
    """SURVEY 8(c) parser-level pins, on the file the reference's TraverseTreeToFile wrote."""
    f = tmp_path / "quadtree_init_cpu.txt"
    f.write_bytes(bytes(gold("ref_project_1024")["quadtree_txt_0"]))
    e = textio.parse_quadtree_file(str(f))
    assert len(e) == 3085
    assert e[0] == (0, -0.119497, 0.119541, -0.119883, 0.11995, 1568.43, [(-1, 0.000603463, -0.00254328)])
    assert e[8] == (5, -0.0896173, -0.0821473, -0.104894, -0.097399, 0.624037, [(608, -0.0827564, -0.0990748)])
    occ = [x[6][0][0] for x in e if x[6]]
    assert len(occ) <= 1793 and sum(1 for o in occ if o == -1) == 773 and sum(1 for o in occ if o >= 0) == 996
    assert sum(1 for x in e if not x[6]) >= 1292


def test_positions_file_round_trip_and_plot(tmp_path):
    """savePositions format (project.cu:855-863) written, read back, and drawn (our plot_2d.py)."""
    from gpu_nbody_simulation_amd import textio
    rng = np.random.default_rng(0)
    frames = [(float(t), rng.uniform(-1, 1, (5, 2))) for t in range(4)]
    path = tmp_path / "positions.txt"
    textio.save_positions(str(path), frames)
    first = open(path).readline()
    assert first.count(" ") == 4 and first.endswith(" \n")            # "t i x y " with the trailing blank
    t, pos = textio.parse_positions_file(str(path))
    assert np.array_equal(t, np.arange(4.0)) and pos.shape == (4, 5, 2)
    np.testing.assert_allclose(pos, np.stack([f[1] for f in frames]), atol=5e-7)   # "%f": 6 decimals
    png = tmp_path / "plot_2d.png"
    assert textio.plot_trajectories(str(path), str(png), max_bodies=3) == 3
    assert png.stat().st_size > 1000
    with open(path, "a") as f:
        f.write("4.000000 0 0.000000 0.000000 \n")                    # an incomplete frame
    with pytest.raises(ValueError):
        textio.parse_positions_file(str(path))
