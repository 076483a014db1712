"""Host mirror of the reference's native transcript code (hekaton_system_amd/transcript.py;
distributed-prover/src/transcript/{mod,rom_transcript,ram_transcript}.rs, coordinator.rs:92-160).  The reference's own
tests for this module compare the native update with the gadget update (rom_transcript.rs:326, ram_transcript.rs:400);
here the native update is compared with the big-merkle job generator's independent formula (sha_circuit.ShaMerkleJob,
whose evals are in turn enforced by the subcircuits' R1CS, tests/test_sha_circuit.py) and with first principles."""
import hashlib
import random

import pytest

from hekaton_system_amd import transcript as tr
from hekaton_system_amd.ark_serialize import ArkCodec, SerializationError
from hekaton_system_amd.cp_groth16 import CURVE_PARAMS
from hekaton_system_amd.sha_circuit import ShaMerkleJob
from hekaton_system_amd.worker import Stage0Request

R = CURVE_PARAMS["bn254"]["r"]


def test_challenges_are_sha256_of_context_and_commitment():
    com = bytes(range(256)) * 3
    ev = tr.RunningEvaluation.new(tr.ROM, com, R)
    assert ev.challenges == tuple(int.from_bytes(hashlib.sha256(t + com).digest(), "little") % R for t in (b"entry_chal", b"tr_chal"))
    ev = tr.RunningEvaluation.new(tr.RAM, com, R)
    tags = (b"entry_chal_1", b"entry_chal_2", b"entry_chal_3", b"tr_chal")
    assert ev.challenges == tuple(int.from_bytes(hashlib.sha256(t + com).digest(), "little") % R for t in tags)
    assert ev.time_ordered_eval == ev.addr_ordered_eval == 1
    # the ROM pair is what aggregation.rom_challenges derives from the same commitment object
    from hekaton_system_amd.aggregation import IppCom, rom_challenges
    from hekaton_system_amd.gt import GtField
    c = IppCom(GtField("bn254"), tuple(range(1, 13)), tuple(range(21, 33)))
    assert tr.RunningEvaluation.new(tr.ROM, c, R).challenges == rom_challenges(c, R)


def test_rom_running_evaluations_match_the_job_generator():
    leaves = [bytes([7 * i + k & 0xff for k in range(64)]) for i in range(4)]
    com = b"some serialized super commitment"
    ev0 = tr.RunningEvaluation.new(tr.ROM, com, R)
    job = ShaMerkleJob("bn254", 8, 1, 4, leaves, *ev0.challenges)
    time_st = [[tr.RomTranscriptEntry(a, v) for a, v in ops] for ops in job.time]
    addr_st = tr.sort_subtraces_by_addr(time_st)
    assert [[(e.addr, e.val) for e in st] for st in addr_st] == job.addr
    out = tr.running_evaluations(tr.ROM, com, R, time_st, addr_st)
    assert [ev.time_ordered_eval for ev, _ in out] == job.time_eval0[1:]
    assert [ev.addr_ordered_eval for ev, _ in out] == job.addr_eval0[1:]
    assert out[-1][0].time_ordered_eval == out[-1][0].addr_ordered_eval          # same multiset
    assert all(last == st[-1] for (_, last), st in zip(out, addr_st))


def test_ram_update_formula_sort_and_permutation_property():
    rnd = random.Random(4)
    n_sub, per = 6, 5
    # a consistent RAM trace: writes then reads of a few addresses with increasing timestamps
    entries, ts = [], 0
    mem = {}
    for _ in range(n_sub * per):
        addr = rnd.randrange(1, 5)
        ts += 1
        if addr not in mem or rnd.random() < 0.4:
            mem[addr] = rnd.randrange(R)
            entries.append(tr.RamTranscriptEntry(addr, mem[addr], ts, False))
        else:
            entries.append(tr.RamTranscriptEntry(addr, mem[addr], ts, True))
    time_st = [entries[k * per:(k + 1) * per] for k in range(n_sub)]
    addr_st = tr.sort_subtraces_by_addr(time_st)
    flat = [e for st in addr_st for e in st]
    assert [e.sort_key() for e in flat] == sorted(e.sort_key() for e in entries)
    assert [len(st) for st in addr_st] == [per] * n_sub
    com = b"ram commitment"
    out = tr.running_evaluations(tr.RAM, com, R, time_st, addr_st)
    c1, c2, c3, t = out[0][0].challenges
    want = 1
    for e in entries:
        want = want * ((t - (e.val + c1 * e.addr + c2 * e.i + c3 * int(e.read))) % R) % R
    assert out[-1][0].time_ordered_eval == want == out[-1][0].addr_ordered_eval
    # a changed value anywhere breaks the equality of the two products
    bad = [list(st) for st in addr_st]
    e = bad[2][1]
    bad[2][1] = tr.RamTranscriptEntry(e.addr, (e.val + 1) % R, e.i, e.read)
    out2 = tr.running_evaluations(tr.RAM, com, R, time_st, bad)
    assert out2[-1][0].time_ordered_eval != out2[-1][0].addr_ordered_eval
    # entry kinds cannot be mixed (the reference panics)
    with pytest.raises(TypeError):
        out[0][0].update_time_ordered(tr.RomTranscriptEntry(1, 2))
    with pytest.raises(RuntimeError):
        tr.RunningEvaluation(tr.RAM, R).update_addr_ordered(entries[0])


def test_wire_round_trips_and_malformed_bytes():
    nb = 32
    e = tr.RamTranscriptEntry(0x1122334455667788, R - 5, 0x80000001, True)
    w = e.to_wire(nb)
    assert len(w) == 8 + nb + 8 + 32 + 1 and w[8 + nb:16 + nb] == (32).to_bytes(8, "little")
    assert w[16 + nb] == 1 and w[16 + nb + 31] == 1 and sum(w[16 + nb:48 + nb]) == 2
    assert tr.RamTranscriptEntry.from_wire(w, 0, nb) == (e, len(w))
    assert tr.RamTranscriptEntry.padding().to_field_elements() == [0, 0, 0, 0]
    bad = bytearray(w)
    bad[16 + nb + 3] = 2
    with pytest.raises(ValueError):
        tr.RamTranscriptEntry.from_wire(bytes(bad), 0, nb)
    for mem, k in ((tr.ROM, 2), (tr.RAM, 4)):
        ev = tr.RunningEvaluation(mem, R, list(range(5, 5 + k)), 77, 88)
        wire = ev.to_wire(nb)
        assert len(wire) == 1 + 2 * nb + 1 + k * nb and wire[0] == (0 if mem == tr.ROM else 1)
        back, off = tr.RunningEvaluation.from_wire(wire, 0, nb, R)
        assert off == len(wire) and back.challenges == ev.challenges and back.time_ordered_eval == 77
        none = tr.RunningEvaluation(mem, R)
        assert tr.RunningEvaluation.from_wire(none.to_wire(nb), 0, nb, R)[0].challenges is None
    with pytest.raises(ValueError):
        tr.RunningEvaluation.from_wire(b"\x02" + bytes(100), 0, nb, R)
    # a Stage0Request with RAM entries through the ark-serialize framing
    codec = ArkCodec("bn254")
    req = Stage0Request(3, [e, tr.RamTranscriptEntry.padding()], [tr.RamTranscriptEntry(1, 2, 3, False)])
    wire = codec.stage0_request_to_wire(req)
    back = codec.stage0_request_from_wire(wire)
    assert back.subcircuit_idx == 3 and back.time_ordered_subtrace == [e, tr.RamTranscriptEntry.padding()]
    assert back.addr_ordered_subtrace == [tr.RamTranscriptEntry(1, 2, 3, False)]
    with pytest.raises(SerializationError):
        codec.stage0_request_from_wire(wire[:8] + (1).to_bytes(8, "little") + b"\x07" + bytes(80))
