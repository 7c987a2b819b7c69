# -*- coding: utf-8 -*-
"""Pipeline API of the hot path (mirror of reference utils/assist.py:66-150, :210-215, :228-244).

Same function names, argument order and return shapes as the reference.  `sc` / `sqlContext` are
accepted and passed through to the returned RDD-like handles, nothing is shuffled or broadcast:
each pipeline is a handful of kernel launches on the MI355X."""
import time
from os import makedirs
from os.path import join

import yaml


def baseliner_clean_data_pipeline(sc, clean_tool, path_rawdata, is_debug, num_partition):
    """parse -> filter -> clean (-> debug subset).  reference utils/assist.py:9-21 (host-side ETL).
    With XMAP_NATIVE_FEED=1 the three steps run in the library's native feeder (csrc/feeder.hip, same semantics) and the
    result is an RDD whose records exist natively: Python tuples only if somebody iterates it."""
    import os
    if os.environ.get("XMAP_NATIVE_FEED") == "1" and not is_debug:
        from xmap.engine import feeder
        period = clean_tool.period
        feed = feeder.Feed.from_file(path_rawdata, period[0], period[-1], clean_tool.label, clean_tool.num_atleast_rating)
        return feeder.FeedRDD(feed, sc)
    dataRDD = sc.textFile(path_rawdata, 30)
    cleanedRDD = clean_tool.clean_data(clean_tool.filter_data(clean_tool.parse_data(dataRDD))).cache()
    if is_debug:
        return sc.parallelize(clean_tool.take_partial_data(cleanedRDD), num_partition).cache()
    return cleanedRDD


def baseliner_split_data_pipeline(sc, split_tool, sourceRDD, targetRDD):
    """(trainRDD, testRDD).  reference utils/assist.py:24-38"""
    overlap_bd = sc.broadcast(split_tool.find_overlap_user(sourceRDD, targetRDD).collect())
    overlap_source, rest_source = split_tool.distinguish_data(overlap_bd, sourceRDD)
    overlap_target, rest_target = split_tool.distinguish_data(overlap_bd, targetRDD)
    trainRDD, testRDD = split_tool.split_data(rest_source, overlap_source, rest_target, overlap_target)
    return trainRDD.cache(), testRDD.cache()


def baseliner_split_multidomain_data_pipeline(sc, split_tool, sourceRDD1, sourceRDD2, targetRDD):
    """(trainRDD1, trainRDD2, testRDD).  reference utils/assist.py:41-63"""
    overlap_bd = sc.broadcast(
        split_tool.find_overlap_user_multidomain(sourceRDD1, sourceRDD2, targetRDD).collect())
    o1, r1 = split_tool.distinguish_data(overlap_bd, sourceRDD1)
    o2, r2 = split_tool.distinguish_data(overlap_bd, sourceRDD2)
    ot, rt = split_tool.distinguish_data(overlap_bd, targetRDD)
    t1, t2, test = split_tool.split_data_multipledomain(r1, o1, r2, o2, rt, ot)
    return t1.cache(), t2.cache(), test.cache()


def baseliner_calculate_sim_pipeline(sc, itemsim_tool, trainRDD):
    """a pipeline to calculate itembased sim.  reference utils/assist.py:66-77
    returns RDD-like[((iid1, iid2), (sim, mutu, frac_mutu, label))]"""
    item2item_simRDD = itemsim_tool.calculate_item2item_sim(trainRDD, None, None)
    # the reference calls .cache() on None for an unknown method (assist.py:75) -> AttributeError
    item2item_simRDD = item2item_simRDD.cache()
    item2item_simRDD.ctx = sc
    return item2item_simRDD


def extender_pipeline(sc, sqlContext, itemsim_tool, extendsim_tool, item2item_simRDD):
    """reference utils/assist.py:80-102.  returns RDD-like[(start_iid, [(end_iid, xsim)*])]"""
    from xmap.engine import session
    from xmap.engine.localrdd import records_of
    from xmap.core.extender import _items_state
    if isinstance(item2item_simRDD, session.SimPairsRDD):
        st, S = item2item_simRDD.state, item2item_simRDD.S
    else:   # generic records (e.g. a re-ordered / filtered copy): ids come from the records
        recs = records_of(item2item_simRDD)
        st = _items_state(sorted({k[0] for k, _ in recs} | {k[1] for k, _ in recs}))
        S = session.sim_from_records(st, recs)
    # lazy: only the per-start candidate counts and top candidates are computed here (what generator_pipeline
    # consumes); the (start, [(end, xsim)*]) lists materialise if the returned RDD is iterated
    E = extendsim_tool.extend(st, S, full=False)
    return session.ExtendedSimRDD(st, E, sc).cache()


def extract_siminfo(sc, classfied_items):
    """(BB_info, NB_info, knn_BB_bd, knn_NB_bd) of the classified items -- reference utils/assist.py:105-133.
    BB_info: (bridge iid, (BB_BB, BB_NB))*, NB_info: (non-bridge iid, (NB_BB, NB_NN))*; the two broadcasts hold
    {iid: {neighbour: (sim, mutu, frac_mutu)}} over both lists of an item (a neighbour listed twice keeps the entry
    of the second list, as dict() does in the reference)."""
    from xmap.engine.localrdd import LocalRDD, records_of
    bb, nb, knn_bb, knn_nb = [], [], {}, {}
    for iid, bridge_lists, other_lists in records_of(classfied_items):
        for lists, info, table in ((bridge_lists, bb, knn_bb), (other_lists, nb, knn_nb)):
            if lists is None:
                continue
            info.append((iid, lists))
            table[iid] = {}
            for lst in lists:
                for entry in lst:
                    table[iid][entry[0]] = tuple(entry[1:])
    ctx = getattr(classfied_items, "ctx", None)
    return LocalRDD(bb, ctx), LocalRDD(nb, ctx), sc.broadcast(knn_bb), sc.broadcast(knn_nb)


def generator_pipeline(privatemap_tool, trainRDD, extended_simRDD, private):
    """a pipeline to private map item.  reference utils/assist.py:136-150
    returns RDD-like[(uid, iid, rating, time)] (all iids target-domain)"""
    from xmap.engine import session
    from xmap.engine.localrdd import records_of
    st = session.train_state(trainRDD)
    if isinstance(extended_simRDD, session.ExtendedSimRDD) and extended_simRDD.state.idt.iids == st.idt.iids:
        E = extended_simRDD.E
    else:
        E = session.ext_from_records(st, records_of(extended_simRDD))
    n_top, choice, mp = privatemap_tool.select(st, E, bool(private))
    G = st.engine.alterego(mp)
    return session.AlterEgoRDD(st, G, getattr(trainRDD, "ctx", None)).cache()


def recommender_calculate_sim_pipeline(sc, cross_sim_tool, alterEgo_profile):
    """similarity over the AlterEgo profile.  reference utils/assist.py:153-177"""
    user_based = cross_sim_tool.build_sthbased_profile(alterEgo_profile, "user").cache()
    item_based = cross_sim_tool.build_sthbased_profile(alterEgo_profile, "item").cache()
    user_based_dict_bd = sc.broadcast(user_based.collectAsMap())
    item_based_dict_bd = sc.broadcast(item_based.collectAsMap())
    user_info_bd = sc.broadcast(cross_sim_tool.get_info(user_based).collectAsMap())
    item_info_bd = sc.broadcast(cross_sim_tool.get_info(item_based).collectAsMap())
    alterEgo_sim = cross_sim_tool.calculate_sim(item_based, user_based, item_info_bd, user_info_bd).cache()
    return user_based, item_based, user_based_dict_bd, item_based_dict_bd, user_info_bd, item_info_bd, alterEgo_sim


def recommender_privacy_pipeline(policy_tool, alterEgo_sim, is_private):
    """neighbour selection + perturbation.  reference utils/assist.py:180-194"""
    if is_private:
        return policy_tool.noise_perturbation(policy_tool.private_neighbor_selection(alterEgo_sim))
    return policy_tool.nonnoise_perturbation(policy_tool.nonprivate_neighbor_selection(alterEgo_sim))


def recommender_prediction_pipeline(recommender_tool, cross_sim_tool, testRDD, simpair_dict_bd,
                                    user_based_dict_bd, item_based_dict_bd, user_info_bd, item_info_bd):
    """MAE string.  reference utils/assist.py:197-207 (the user-based branch exists only in the reference's egg)"""
    if "user" in cross_sim_tool.method:
        predicted = recommender_tool.user_based_recommendation(
            testRDD, user_based_dict_bd, simpair_dict_bd, user_info_bd)
    else:
        predicted = recommender_tool.item_based_recommendation(
            testRDD, item_based_dict_bd, simpair_dict_bd, item_info_bd)
    return recommender_tool.calculate_mae(predicted)


def map_to_dict(rdd):
    """{source item: target item} -- reference utils/assist.py:210-215 (last writer wins)."""
    return dict((line[1], line[0]) for line in rdd.collect())


def load_parameter(path):
    """reference utils/assist.py:228-231 (explicit Loader: PyYAML >= 6 requires one)."""
    with open(path, 'rb') as f:
        return yaml.load(f, Loader=yaml.SafeLoader)


def write_to_disk(results, out_dict, path):
    """reference utils/assist.py:234-244"""
    timestamp = str(int(time.time()))
    out_folder = join(path, "runs", timestamp)
    makedirs(out_folder)
    out_dict['result'] = results
    with open(join(out_folder, "info.yaml"), 'w') as yaml_file:
        yaml_file.write(yaml.dump(out_dict, default_flow_style=False))
