"""The multi-agent env rules of the oracle against the reference's OWN classes (fixture: tests/golden/marl_rules_v0.json.gz,
made by oracle/gen_marl_rules.py in the build container): the oracle replays the episode batch of the generator, must
reproduce its event stream, and its bookkeeping must equal what StayTimeManager / TollGateObservation /
ParkingLotSpawnManager answered on those events."""
import gzip
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def golden():
    with gzip.open(os.path.join(ROOT, "tests", "golden", "marl_rules_v0.json.gz"), "rt") as fh:
        return json.load(fh)


@pytest.mark.timeout(900)
def test_tollgate_stay_time_and_toll_observation_match_the_reference_classes(golden):
    from oracle import gen_marl_rules
    run = gen_marl_rules.toll_run()  # oracle only (the reference is not needed, and not available, at test time)
    g = golden["toll"]
    assert run["events"] == g["events"], "the oracle run no longer reproduces the event stream the fixture was made from"
    ref = g["reference"]
    n_entry = n_exit = n_rows = n_in_toll = n_long = n_stay_done = 0
    for t, (ev_t, bk_t, rw_t) in enumerate(zip(run["events"], run["book"], run["rows"])):
        for e, ev in enumerate(ev_t):
            rows = {r[0]: r for r in rw_t[e]}
            for aid in ref["stay_done"][t][e]:  # crossed the plaza in < min_pass_steps: terminated with out_of_road
                if aid in rows:
                    assert rows[aid][3] == 1 and rows[aid][4] == 1, (t, e, aid)
                    n_stay_done += 1
            if ev["reset"]:
                continue
            for (aid, blk), (php, phi, plp, pli), (entry, exit_, last), (f0, f1, cnt) in zip(ev["active"], bk_t[e], ref["stay"][t][e],
                                                                                        ref["obs_replay"][t][e]):
                # StayTimeManager.record: entry / exit step and last block of every active agent
                assert phi == entry and plp == exit_ and chr(int(pli)) == last, (t, e, aid, (phi, plp, pli), (entry, exit_, last))
                # TollGateObservation: the counter, and (for agents that reported this step) the two floats of their row
                assert php == cnt, (t, e, aid, php, cnt)
                if aid in rows:
                    assert rows[aid][1] == f0 and rows[aid][2] == f1, (t, e, aid, rows[aid], (f0, f1))
                    n_rows += 1
                    n_in_toll += f0 > 0
                    n_long += f1 > 0
                n_entry += entry >= 0
                n_exit += exit_ >= 0
    print("tollgate rules: rows", n_rows, "in plaza", n_in_toll, "stayed long", n_long, "entry records", n_entry, "exit records", n_exit,
          "stay-time terminations", n_stay_done)
    assert n_rows > 15000 and n_in_toll > 300 and n_long > 50 and n_entry > 400 and n_exit > 80 and n_stay_done >= 5
    # the counter / floats rule on synthetic block sequences (marl_tollgate.py:84-96): [in plaza, in_toll_time > min_pass_steps]
    for case in ref["obs_cases"]:
        cnt = 0
        for blk, (f0, f1, c) in zip(case["seq"], case["out"]):
            cnt += blk == "$"
            assert c == cnt and f0 == float(blk == "$") and f1 == float(blk == "$" and cnt > 30)


@pytest.mark.timeout(900)
def test_parking_space_pool_matches_the_reference_spawn_manager(golden):
    from oracle import gen_marl_rules
    run = gen_marl_rules.parking_run()
    g = golden["parking"]
    assert run["events"] == g["events"] and run["n_spaces"] == g["n_spaces"]
    n_take = n_done_held = n_check = 0
    for ev_e, ref_e in zip(run["events"], g["reference"]):
        for ev, (ref_free, ref_holders) in zip(ev_e, ref_e):
            if ev["kind"] in ("reset", "check"):  # free spaces + who holds one, after the reference processed the same events
                assert ev["free"] == ref_free and ev["holders"] == ref_holders, (ev, ref_free, ref_holders)
                assert ev["free"] + len(ev["holders"]) == g["n_spaces"]
                n_check += 1
            n_take += ev["kind"] == "spawn" and ev["takes"]
            n_done_held += ev["kind"] == "done" and ev["held"]
    print("parking rules: checks", n_check, "spaces handed out on respawn", n_take, "spaces returned", n_done_held)
    assert n_check > 1000 and n_take > 20 and n_done_held > 20


def test_one_respawn_per_step_matches_the_reference_respawn_loop(golden):
    """How many agents one env.step respawns.  The fixture holds what the reference's OWN SpawnManager.get_available_respawn_places +
    MultiAgentPGDrive._respawn_vehicles do on 24 occupancy patterns of eight places (oracle/gen_marl_rules.py::respawn_reference):
    a free place is offered once per frame, one offered place is taken per call -- ONE newcomer per step when any place is free,
    none from a second call in the same frame, the next one in the next frame.  (Rounds 2 - 4 filled every free place in one step;
    the re-stated reference invariants of tests/test_marl_invariants_gpu.py found it.)  The oracle is held to the same rule: with
    ten seats emptied at once it brings back exactly one agent per step, each into a free place."""
    import numpy as np
    from oracle import orc
    from pgdrive_amd import _abi
    from tests import util
    cases = golden["respawn"]
    assert len(cases) >= 20
    for c in cases:
        n_free = c["places"] - len(c["occupied"])
        assert c["newcomers"] == min(1, n_free) and c["newcomers_second_call_same_frame"] == 0
        assert c["newcomers_next_frame"] == min(1, n_free)  # (the stand-in vehicles occupy nothing: the same places are free again)
        assert c["offered_in_first_call"] == [p for p in range(c["places"]) if p not in c["occupied"]]
    d, mb, sb = util.make_marl_banks(num_agents=40, n_variants=2, kind="roundabout")
    n = 6
    cfg = util.marl_config(n, sb, horizon=1000, delay_done=0)
    ora = orc.Oracle(cfg, mb, sb)
    ora.reset(np.arange(n) % 2)
    f, i, ei = ora.get_state()
    i[_abi.SI["STATUS"], :, :10] = _abi.ST_EMPTY  # ten agents leave at once: ten seats free, alive = 30 < 40
    ora.set_state(f, i, ei)
    act = np.zeros((n, 40, 2), np.float32)
    act[..., 1] = 0.5  # everybody drives off: the road starts clear one after the other
    back = np.zeros(n, int)
    for t in range(60):
        f0, i0, _ = ora.get_state()
        obs, rew, done, flags = ora.step(act)
        new = (flags & _abi.F_NEW) != 0
        assert (new.sum(axis=1) <= 1).all(), "more than one newcomer in one env step"
        f1, i1, _ = ora.get_state()
        for e in range(n):
            for a in np.nonzero(new[e])[0]:  # the newcomer stands clear of everybody who was there before it came
                others = [(f0[_abi.SF["X"], e, b], f0[_abi.SF["Y"], e, b]) for b in range(40)
                          if b != a and i0[_abi.SI["STATUS"], e, b] in (_abi.ST_ACTIVE, _abi.ST_DYING)]
                dist = min(np.hypot(f1[_abi.SF["X"], e, a] - x, f1[_abi.SF["Y"], e, a] - y) for x, y in others)
                assert dist > 3.0
        back += new.sum(axis=1)
    assert (back >= 6).all() and (back <= 60).all()  # one per step, and only while some place is free
    ora.close()
