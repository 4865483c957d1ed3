"""The loss-term lists of the reference's ``engine/organize_loss.py:control_loss`` (the product's ``engine.py`` is the launch
orchestration, so the lists live beside the trainer that reads them)."""


def control_loss(Train_stage):
    """-> (name_fs_list, name_recon_list, name_geo_list, name_prop_list, name_TDA_list), as engine/organize_loss.py:1-27"""
    stages = {
        'PoseNet_only': (['Rot1', 'Rot2', 'Rot1_cos', 'Rot2_cos', 'Rot_regular', 'Tran', 'Size', 'R_con'], ['Per_point', 'Point_voting'],
                         ['Geo_point'], ['Prop_pm', 'Prop_sym'], []),
        'FSNet_only': (['Rot1', 'Rot2', 'Tran', 'Size', 'Recon'], [], [], [], []),
        'TDA': ([], [], [], [], ['Rot1', 'Rot2', 'Rot1_cos', 'Rot2_cos', 'Rot_regular', 'Tran', 'Size', 'R_con', 'TDA_h1', 'TDA_h2',
                                 'TDA_h1_cate', 'TDA_h2_cate', 'Prop_sym', 'R_DCD_cate_pred']),
    }
    if Train_stage not in stages:
        raise NotImplementedError
    return tuple(list(l) for l in stages[Train_stage])
