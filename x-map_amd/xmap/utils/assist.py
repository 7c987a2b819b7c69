# -*- coding: utf-8 -*-
"""Pipeline API of the hot path (mirror of reference utils/assist.py:66-150, :210-215, :228-244).

Same function names, argument order and return shapes as the reference.  `sc` / `sqlContext` are
accepted and passed through to the returned RDD-like handles, nothing is shuffled or broadcast:
each pipeline is a handful of kernel launches on the MI355X."""
import time
from os import makedirs
from os.path import join

import yaml


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
    E = extendsim_tool.extend(st, S, full=True)
    return session.ExtendedSimRDD(st, E, sc).cache()


def extract_siminfo(sc, classfied_items):
    """reference utils/assist.py:105-133 (host-side; the engine keeps these tables in HBM instead)."""
    BB_info = classfied_items.map(lambda line: (line[0], line[1])).filter(lambda line: line[1] is not None)
    NB_info = classfied_items.map(lambda line: (line[0], line[2])).filter(lambda line: line[1] is not None)
    BB_items_knn = BB_info.map(
        lambda line: (line[0], dict((l[0], l[1:]) for l in line[1][0] + line[1][1]))).collectAsMap()
    NB_items_knn = NB_info.map(
        lambda line: (line[0], dict((l[0], l[1:]) for l in line[1][0] + line[1][1]))).collectAsMap()
    return BB_info, NB_info, sc.broadcast(BB_items_knn), sc.broadcast(NB_items_knn)


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
