"""
`RunTHetA` command line on the GPU search: the reference's main() / run_fixed_N() flow
(python/RunTHetA.py:278-509) with the same flags and the same .withBounds / .results / .RunN3.bash /
.BEST.results files, minus clustering, the BAF model and plotting (out of scope).

    python -m theta_amd.RunTHetA example/Example.intervals -n 2 -k 3 -d out/
"""
import os
import sys

from . import search as _search
from .CalcAllC import calc_all_c_2, calc_all_c_3, calc_all_c_3_multi_event
from .DataTools import (calculate_bounds_heuristic, calculate_bounds_normal_heuristic, determine_frac_copy_num,
                        reverse_sort_list, set_total_read_counts, sort_by_sorted_index, sort_r)
from .FileIO import (parse_arguments, read_interval_file, read_results_file, write_out_bounds, write_out_N3_script,
                     write_out_result)
from .ModelSelection import ModelSelection
from .SelectIntervals import select_intervals_n2, select_intervals_n3
from .TimeEstimate import time_estimate
from .search import do_optimization, do_optimization_single, find_mins


def best_near_max_contamination(best, max_normal):
    """RunTHetA.py:222-225."""
    return any(abs(max_normal - mu[0]) < .01 for _, mu, _, _ in best)


def run_fixed_N(n, args, intervals, resultsfile=None):
    """RunTHetA.py:298-509."""
    (filename, results, N, k, tau, directory, prefix, max_normal, bound_heuristic, normal_bound_heuristic, heuristic_lb,
     heuristic_ub, num_processes, bounds_only, multi_event, force, get_values, choose_intervals, num_intervals,
     read_depth_file, graph_format, runBAF, ratio_dev, min_frac, tumorfile, normalfile, noClustering) = args
    lengths, tumorCounts, normCounts, m, upper_bounds, lower_bounds = intervals
    _search.pre = prefix          # RunTHetA.py:307-308: the --GET_VALUES dump goes to <prefix>.likelihoods in the WORKING directory, not -d
    if tumorfile is not None or normalfile is not None or runBAF:
        print("NOTE: SNP files / the BAF model / interval clustering are not part of this implementation; continuing without them.")

    frac = determine_frac_copy_num(normCounts, tumorCounts, lengths, ratio_dev)
    print("Frac with potential copy numbers:", frac)
    if frac < min_frac:
        print("ERROR: This sample does not have enough large copy number aberrations to be a good candidate for tumor "
              "composition estimation using THetA.  See --RATIO_DEVIATION and --MIN_FRAC flags to modify how the "
              "potential presence of large copy number aberrations is determined.  Exiting...")
        sys.exit(1)

    order = None
    if choose_intervals:
        print("Selecting intervals...")
        allTumor, allNormal = tumorCounts, normCounts
        if n == 2:
            if lower_bounds is None or upper_bounds is None:
                order, lengths, tumorCounts, normCounts = select_intervals_n2(lengths, tumorCounts, normCounts, m, k, force,
                                                                              num_intervals)
                upper_bounds = lower_bounds = None
            else:
                order, lengths, tumorCounts, normCounts, lower_bounds, upper_bounds = select_intervals_n2(
                    lengths, tumorCounts, normCounts, m, k, force, num_intervals, lower_bounds, upper_bounds)
        else:
            if resultsfile is None:
                print("ERROR: No results file supplied. Unable to automatically select intervals for n=3 without results "
                      "of n=2 analysis. See --RESULTS flag, or --NO_INTERVAL_SELECTION to disable interval selection. "
                      "Exiting...")
                sys.exit(1)
            copy = read_results_file(resultsfile)
            order, lengths, tumorCounts, normCounts, upper_bounds, lower_bounds, copy = select_intervals_n3(
                lengths, tumorCounts, normCounts, m, upper_bounds, lower_bounds, copy, tau, force, num_intervals)
        m = len(order)

    set_total_read_counts(sum(tumorCounts), sum(normCounts))
    print("Preprocessing data...")
    r, rN, sorted_index = sort_r(normCounts, tumorCounts)

    if normal_bound_heuristic is not False:
        upper_bounds, lower_bounds = calculate_bounds_normal_heuristic(normal_bound_heuristic, heuristic_lb, heuristic_ub,
                                                                       r, rN, m, k)
    elif bound_heuristic is not False or (upper_bounds is None and lower_bounds is None):
        if bound_heuristic is False:
            bound_heuristic = 0.5
        upper_bounds, lower_bounds = calculate_bounds_heuristic(float(bound_heuristic), r, rN, m, tau, k)
    else:
        # bounds from the file are strings; the reference only survives that on the interval-selection path
        upper_bounds = [int(v) for v in sort_by_sorted_index(upper_bounds, sorted_index)]
        lower_bounds = [int(v) for v in sort_by_sorted_index(lower_bounds, sorted_index)]

    ub_out = reverse_sort_list(upper_bounds, sorted_index)
    lb_out = reverse_sort_list(lower_bounds, sorted_index)
    boundsfile = write_out_bounds(directory, prefix, filename, ub_out, lb_out, n, order if choose_intervals else None)
    if bounds_only:
        sys.exit(0)

    time_estimate(n, m, k, tau, lower_bounds, upper_bounds, r, rN, max_normal, sorted_index, num_processes, multi_event, force)
    print("Performing optimization...")
    if num_processes == 1:
        best = do_optimization_single(n, m, k, tau, lower_bounds, upper_bounds, r, rN, max_normal, sorted_index,
                                      multi_event, get_values)
    else:
        best = do_optimization(n, m, k, tau, lower_bounds, upper_bounds, r, rN, max_normal, sorted_index, num_processes,
                               multi_event, get_values)
    if best == []:
        print("ERROR: Maximum Likelihood Solution not found within given bounds.")
        sys.exit(1)
    rep = _search.last_report
    print("\tSearched %d candidate matrices in %.2f s on %s (%d finalists)%s" % (
        rep.candidates, rep.seconds, "the GPU" if getattr(rep, "gpus", 1) == 1 else "%d GPU ranks" % rep.gpus, rep.finalists,
        "" if rep.certificate_complete in (True, None) else "; suspect list overflowed: rerun with a tighter rank range (see DESIGN.md section 5)"))
    mix = getattr(rep, "mix", None)
    if mix and "gave_up" not in mix:
        # a space no walk finishes, searched whole over the mixture space: what that covers
        print("\tWhole space by branch and bound over the mixture space: %d boxes, %d matrices listed; every finite outcome within the tie "
              "window is among the %d records (rank-deficient matrices: nothing finite below %.3f); NaN outcomes are not listed." % (
                  mix.get("boxes_tested", 0), mix.get("listed", 0), mix.get("records", 0), mix.get("rank_deficient_bound") or float("nan")))

    if getattr(rep, "libm_pow_matches", None) is False:
        print("NOTE: this host's libm rounds pow(x, 2) differently from the one the n=3 kernels restate (glibc >= 2.28, x86-64 with "
              "FMA): the reference run HERE would report other values for rank-deficient candidate matrices (INTEGRATION.md section 5).")
    if n == 2 and best_near_max_contamination(best, max_normal):
        print("WARNING: At least one of the top solutions is near the upper bound on normal contamination. Further "
              "analysis may required (see --MAX_NORMAL and the expected copy number --TAU).")
    r = reverse_sort_list(r, sorted_index)
    rN = reverse_sort_list(rN, sorted_index)

    if choose_intervals:
        if n == 2:
            best = calc_all_c_2(best, r, rN, allTumor, allNormal, order)
        elif n == 3 and not multi_event:
            best = calc_all_c_3(best, r, rN, allTumor, allNormal, order)
        else:
            best = calc_all_c_3_multi_event(best, r, rN, allTumor, allNormal, order)
        best = find_mins(best)          # RunTHetA.py:476: keep the solution(s) with the overall minimum

    resultsfile = write_out_result(directory, prefix, best, n)
    if n == 2:
        write_out_N3_script(directory, prefix, filename)
    return resultsfile, boundsfile


def main(argv=None):
    """RunTHetA.py:278-295."""
    args = parse_arguments(argv)
    print("Reading in query file...")
    intervals = read_interval_file(args[0])
    if args[2] is not None:
        run_fixed_N(args[2], args, intervals, args[1])
    else:
        resultsfile2, boundsfile2 = run_fixed_N(2, args, intervals)
        intervals = read_interval_file(boundsfile2)
        # (the same args for both stages, like the reference: its "20 intervals for n=3" default is applied by the parser
        # only when -n 3 is given, FileIO.py:170 -- the two-stage run selects up to NUM_INTERVALS = 100 for n=3 as well)
        resultsfile3, boundsfile3 = run_fixed_N(3, args, intervals, resultsfile2)
        ModelSelection(args[0], resultsfile2, resultsfile3)


if __name__ == "__main__":
    main()
